"""Which host calls of a ball-domain outer iteration wait for the GPU?  torch's sync debug mode warns at every synchronising
call; the warnings of the LAST iterations of a short run (everything warmed up) are counted per source line."""
import os, sys, warnings, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import configs.Ex4_3_funcs as P
from src.training import NODE_WAN_solver
name = sys.argv[1] if len(sys.argv) > 1 else 'NSphere_THourglass'
params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 10, 'N_t': 20, 'N_r': 8192, 'N_b': 8192, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 3, 'domain': name}
torch.manual_seed(0); np.random.seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './',
                    func_u_sol=getattr(P, 'func_u_sol', None), p=2)
os.makedirs('/tmp/c5', exist_ok=True); os.chdir('/tmp/c5')
S.train()
S.iterations = 4
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter('always')
    import traceback
    seen = collections.Counter()
    orig = warnings.showwarning
    def note(message, category, filename, lineno, file=None, line=None):
        stack = [f for f in traceback.extract_stack() if '/root/repo' in f.filename or 'xnode' in f.filename or 'configs' in f.filename][-4:]
        seen[' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(stack))] += 1
    warnings.showwarning = note
    warnings.simplefilter('always')
    S.train()
torch.cuda.set_sync_debug_mode('default')
for k, v in seen.most_common(40):
    print('%4d  %s' % (v, k))
