"""a short train() run at the headline configuration for a kernel trace (tools/train_trace.sh): 5 + 60 outer iterations"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=5), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/tt', exist_ok=True); os.chdir('/tmp/tt')
S.train()
S.iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.cuda.synchronize(); t0 = time.perf_counter(); S.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('%.3f ms per outer iteration' % (1e3 * dt / S.iterations))
