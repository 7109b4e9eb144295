"""TIMING EXPERIMENT: one HIP graph per g,g,d cycle in which the test network of the NEXT sub-step is launched as soon as this
sub-step's sweep B has been issued (phi does not change in a generator sub-step), on a stream of its own, so that it runs beside
the reduction, the update and the launch gaps that end the sub-step -- against one graph per cycle without that, and one graph per
sub-step.  The early launch writes the SAME output buffers (same phi, same sample: same values) -- not a production schedule (it
would need a second output set), a bound on what one could gain:   python tools/cycle_prefetch.py [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
from xnode_wan_pde_solver_amd import kernels as KN
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda')
torch.manual_seed(0)
S = NODE_WAN_solver(workload_params(20, 4096, 4096, 32), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, dev, './',
                    func_u_sol=P.func_u_sol, p=2)
eng, s = S.engine, S.setup
domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, dev)[0]
G = eng.load_group(du, dv, bd, domain)
for _ in range(4):
    eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
torch.cuda.synchronize()

real_launch, real_bwd = eng._launch_test_net_here, KN.ode_bwd_multi
state = dict(pref=None, nxt=None)


def launch(G_, blocks=None):
    if state['pref'] is not None:                     # this sub-step's test network was launched early: wait for it here
        torch.cuda.current_stream().wait_event(state['pref'])
        state['pref'] = None
        return
    real_launch(G_, blocks)


def bwd(jobs, *a, **kw):
    out = real_bwd(jobs, *a, **kw)
    if state['nxt'] is not None and kw.get('want_params') and len(jobs) == 1 and 'weak' in (jobs[0].get('res') or {}):
        kind, blocks = state['nxt']                   # sweep B has just been issued: the NEXT sub-step's test network, now
        e = torch.cuda.current_stream().record_event()
        pf = eng.streams[0]
        pf.wait_event(e)
        with torch.cuda.stream(pf):
            keep = getattr(G, 'vact_valid', False)
            G.vact_valid = kind == 'disc'
            real_launch(G, blocks)
            G.vact_valid = keep
            state['pref'] = pf.record_event()
    return out


def cycle_body(prefetch):
    state['pref'] = None
    eng._v_fresh(G)
    state['nxt'] = ('gen', eng.v_blocks) if prefetch else None
    eng._gen_all(G)
    eng._v_fresh(G)
    state['nxt'] = ('disc', eng.v_blocks_disc) if prefetch else None
    eng._gen_all(G)
    state['nxt'] = None
    eng._v_fresh(G, store=True)
    eng._phi_version += 1
    eng._disc_all(G)


graphs = {}
for prefetch in (False, True):
    eng._launch_test_net_here, KN.ode_bwd_multi = (launch, bwd) if prefetch else (real_launch, real_bwd)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=eng._capture_stream(), capture_error_mode='thread_local'):
        cycle_body(prefetch)
    graphs[prefetch] = g
eng._launch_test_net_here, KN.ode_bwd_multi = real_launch, real_bwd
for rep in range(3):
    line = []
    for prefetch in (False, True):
        g = graphs[prefetch]
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize(); t1 = 1e3 * (time.perf_counter() - t0) / n
        line.append('cycle graph%s %.4f ms (%.0f sub-steps/s)' % (' + early test network' if prefetch else '', t1, 3e3 / t1))
    for _ in range(3):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize(); t3 = 1e3 * (time.perf_counter() - t0) / n
    print('   '.join(line) + '   one graph per sub-step %.4f ms (%.0f sub-steps/s)' % (t3, 3e3 / t3), flush=True)
print('finite:', bool(torch.isfinite(eng.theta.data).all() and torch.isfinite(eng.phi.data).all()))
