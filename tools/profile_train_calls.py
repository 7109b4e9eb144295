import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=5), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/pt', exist_ok=True); os.chdir('/tmp/pt')
S.train(); S.iterations = 25; S.train()
pr = cProfile.Profile(); pr.enable()
for _ in range(8): S.train()
torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(25); print(st.getvalue()[:5000])
