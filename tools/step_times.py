"""time generator-only and discriminator-only loops of the headline workload (scheduling experiments)
usage: python tools/step_times.py [n]      env: XW_V_BLOCKS, XW_STREAMS, XW_GRAPHS, XW_PREFETCH_V, ..."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda')
torch.manual_seed(0)
S = NODE_WAN_solver(workload_params(20, 4096, 4096, 32), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, dev, './',
                    func_u_sol=P.func_u_sol, p=2)
eng, s = S.engine, S.setup
domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, dev)[0]
G = eng.load_group(du, dv, bd, domain)
out = {}
for name, fn in (('gen', eng.generator_step), ('disc', eng.discriminator_step)):
    for _ in range(5):
        fn(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn(G)
    torch.cuda.synchronize(); out[name] = 1e3 * (time.perf_counter() - t0) / n
for _ in range(6):
    eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n // 3):
    eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
torch.cuda.synchronize(); cycle_real = 1e3 * (time.perf_counter() - t0) / (n // 3)
print('real cycle %.4f ms -> %.1f steps/s' % (cycle_real, 3e3 / cycle_real))
print('v_blocks %s  gen %.4f ms  disc %.4f ms  cycle(g,g,d) %.4f ms -> %.1f steps/s' % (
    os.environ.get('XW_V_BLOCKS', '0'), out['gen'], out['disc'], 2 * out['gen'] + out['disc'], 3e3 / (2 * out['gen'] + out['disc'])))
