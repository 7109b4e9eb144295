import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 20), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
if len(sys.argv) > 2:
    S.tabulate_on_host = sys.argv[2] == 'host'
    S.device_sampling = sys.argv[2] == 'devsample'
os.makedirs('/tmp/pt', exist_ok=True); os.chdir('/tmp/pt')
S.iterations = 3; S.train()          # warm-up (graph capture etc.)
S.iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pr = cProfile.Profile(); t0 = time.time(); pr.enable(); S.train(); torch.cuda.synchronize(); pr.disable(); dt = time.time() - t0
print('outer iterations %d  wall %.3f s  -> %.1f ms / outer iteration' % (S.iterations, dt, 1e3 * dt / S.iterations))
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(28); print(st.getvalue()[:5500])
