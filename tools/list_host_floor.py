"""Ball domains: what would an outer iteration take if the sampling thread cost nothing?  After a warm-up the loader hands out
samples drawn before (timing only: the run trains on recycled samples), so the main thread runs without a busy second thread
beside it.  Compare with tools/train_cfg5.py: the difference is what the draws cost the MAIN thread (interpreter lock)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import configs.Ex4_3_funcs as P
from src.training import NODE_WAN_solver
name = sys.argv[1] if len(sys.argv) > 1 else 'NSphere_THourglass'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 10, 'N_t': 20, 'N_r': 8192, 'N_b': 8192, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 2, 'domain': name}
torch.manual_seed(0); np.random.seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './',
                    func_u_sol=getattr(P, 'func_u_sol', None), p=2)
os.makedirs('/tmp/c5', exist_ok=True); os.chdir('/tmp/c5')
real_loader, real_domain = S._loader, S._new_domain
bank, calls = [], [0]
def loader(domain):
    calls[0] += 1
    if len(bank) < 8:
        ld = real_loader(domain); bank.append((domain, ld)); return ld
    return bank[calls[0] % 8][1]
def new_domain():
    if len(bank) < 8:
        return real_domain()
    return bank[(calls[0] + 1) % 8][0]
S._loader, S._new_domain = loader, new_domain
S.iterations = 6
S.train()
S.iterations = iters
S.__dict__.pop('_list_phase_seconds', None)
torch.cuda.synchronize(); t0 = time.time(); S.train(); torch.cuda.synchronize(); dt = time.time() - t0
print('%s with recycled samples: %.1f ms per outer iteration' % (name, 1e3 * dt / iters))
ph = getattr(S, '_list_phase_seconds', None)
if ph:
    print('  host ms per outer iteration by phase: ' + ', '.join('%s %.2f' % (k_, 1e3 * v / iters) for k_, v in ph.items()))
