"""BASELINE config 5 (d=10 time-varying ball, N_r=N_b=8192, N_t=20, Ex4_3 functions): wall time per outer iteration"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import configs.Ex4_3_funcs as P
from src.training import NODE_WAN_solver
name = sys.argv[1] if len(sys.argv) > 1 else 'NSphere_TCone'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 10, 'N_t': 20, 'N_r': 8192, 'N_b': 8192, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 2, 'domain': name}
torch.manual_seed(0); np.random.seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './',
                    func_u_sol=getattr(P, 'func_u_sol', None), p=2)
os.makedirs('/tmp/c5', exist_ok=True); os.chdir('/tmp/c5')
S.train()
S.iterations = iters
S.__dict__.pop('_list_phase_seconds', None)
torch.cuda.synchronize(); t0 = time.time(); S.train(); torch.cuda.synchronize(); dt = time.time() - t0
drew = ('samples drawn by the sampling process (XW_SAMPLER_PROCESS=0: the helper thread)' if getattr(S, '_sampler_proc', None) is not None
        else 'the sampling thread drew for %.1f ms of each (two samples: the diagnostic\'s and the next iteration\'s)' % (1e3 * S._sampler_seconds / iters))
print('%s: %d outer iterations, %.1f ms each; groups per sample: %d; %s' % (name, iters, 1e3 * dt / iters, len(S._group_cache), drew))
ph = getattr(S, '_list_phase_seconds', None)
if ph:
    print('  host ms per outer iteration by phase: ' + ', '.join('%s %.2f' % (k_, 1e3 * v / iters) for k_, v in ph.items()))
if len(sys.argv) > 3:
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable(); S.iterations = 10; S.train(); torch.cuda.synchronize(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats(sys.argv[4] if len(sys.argv) > 4 else 'cumulative').print_stats(45); print(st.getvalue()[:9000])
