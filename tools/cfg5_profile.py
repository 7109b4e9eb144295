"""cProfile of the generator / discriminator sub-iterations over the groups of one cone sample (BASELINE config 5: d = 10,
N_r = N_b = 8192, N_t = 20): where the host time of ~540 eager launches per outer iteration goes"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import configs.Ex4_3_funcs as P
from src.training import NODE_WAN_solver
name = sys.argv[1] if len(sys.argv) > 1 else 'NSphere_TCone'
params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 10, 'N_t': 20, 'N_r': 8192, 'N_b': 8192, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 3, 'domain': name}
torch.manual_seed(0); np.random.seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './',
                    func_u_sol=getattr(P, 'func_u_sol', None), p=2)
os.makedirs('/tmp/c5', exist_ok=True); os.chdir('/tmp/c5')
S.train()
torch.set_num_threads(4)
S.iterations = 10
torch.cuda.synchronize(); t0 = time.perf_counter(); S.train(); torch.cuda.synchronize()
print('%s: %.2f ms per outer iteration' % (name, 1e3 * (time.perf_counter() - t0) / 10))
pr = cProfile.Profile(); pr.enable(); S.train(); torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(30); print(st.getvalue()[:6000])
