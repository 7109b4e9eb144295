"""memory held after building and training several solvers in one process (the kept graphs' pools)"""
import os, sys, gc
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
os.makedirs('/tmp/mm', exist_ok=True); os.chdir('/tmp/mm')
for i in range(6):
    torch.manual_seed(i)
    S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=4), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                        torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
    S.train(); torch.cuda.synchronize()
    del S; gc.collect(); torch.cuda.empty_cache()
    print('after solver %d: allocated %.0f MB, reserved %.0f MB' % (i, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)
