"""GPU time of one refill of the headline group (Engine.refill_compact: static inputs + one graph replay) and of the captured
L^p diagnostic, against the eager load_group / L_norm they replace:   python tools/refill_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from xnode_wan_pde_solver_amd import sampling
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=2), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
eng, dev = S.engine, S.device
dom = S._new_domain(); pts = S._loader(dom).pin(); comp = pts.compact()
td = comp[0].to(dev)
mk = lambda x: sampling._paths(td, x.to(dev))
def timed(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
G0 = eng.load_group(mk(comp[1]), mk(comp[2]), mk(comp[3]), dom, shared_grid_t0=float(comp[0][0]))
print('load_group, eager (host-bound)         %.3f ms' % timed(lambda: eng.load_group(mk(comp[1]), mk(comp[2]), mk(comp[3]), dom, into=G0, shared_grid_t0=float(comp[0][0])), 50))
for streams in (True, False):
    eng.use_streams = streams
    G = eng.load_group(mk(comp[1]), mk(comp[2]), mk(comp[3]), dom, shared_grid_t0=float(comp[0][0]))
    G.persistent = True
    print('refill_compact, branches=%-5s          %.3f ms' % (streams, timed(lambda: eng.refill_compact(G, comp, dom))))
eng.use_streams = True
S._lean_off = True
print('diagnostic, eager                      %.3f ms' % timed(lambda: S._l_norm(pts, dom.V(), as_tensor=True), 50))
print('diagnostic, one replay                 %.3f ms' % timed(lambda: S._l_norm_replayed(G, pts, dom)))
# GPU-side duration of the replays when the host is ahead (behind 2 ms of queued work: events around the replay)
def gpu_side(fn, n=30):
    tot = 0.0
    x = torch.empty(64 << 20, device=dev)
    for _ in range(n):
        for _ in range(40):
            x.zero_()                                   # ~2 ms of queued work: the launches below are issued long before they run
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n
for streams in (True, False):
    eng.use_streams = streams
    Gs = eng.load_group(mk(comp[1]), mk(comp[2]), mk(comp[3]), dom, shared_grid_t0=float(comp[0][0]))
    Gs.persistent = True
    eng.refill_compact(Gs, comp, dom)
    print('GPU side: refill_compact, branches=%-5s %.3f ms' % (streams, gpu_side(lambda: eng.refill_compact(Gs, comp, dom))))
eng.use_streams = True
print('GPU side: diagnostic replay             %.3f ms' % gpu_side(lambda: S._l_norm_replayed(G, pts, dom)))
