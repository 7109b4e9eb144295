"""one HIP graph for a whole g,g,d cycle against one graph per sub-step (the hop between two graph launches: end-of-graph signal,
next launch's first dispatch): python tools/cycle_graph.py [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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


def cycle_body():
    for _ in range(2):
        eng._v_fresh(G)
        eng._gen_all(G)
    eng._v_fresh(G, store=True)
    eng._phi_version += 1
    eng._disc_all(G)


g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=eng._capture_stream(), capture_error_mode='thread_local'):
    cycle_body()
for rep in range(3):
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); t1 = 1e3 * (time.perf_counter() - t0) / n
    for _ in range(3):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize(); t3 = 1e3 * (time.perf_counter() - t0) / n
    print('one graph per cycle %.4f ms (%.0f sub-steps/s)   one graph per sub-step %.4f ms (%.0f sub-steps/s)' % (t1, 3e3 / t1, t3, 3e3 / t3), flush=True)
