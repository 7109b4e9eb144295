import os, sys, time, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import configs.Ex4_1_funcs as P
from src.training import NODE_WAN_solver
params = {'alpha': 1e8, 'u_layers': 8, 'u_hidden_dim': 48, 'u_hidden_hidden_dim': 16, 'v_layers': 9, 'v_hidden_dim': 100,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 5, 'N_t': 16, 'N_r': 256, 'N_b': 64, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 20, 'domain': 'Hypercube'}
os.makedirs('/tmp/wt', exist_ok=True); os.chdir('/tmp/wt')
torch.manual_seed(0)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
print([str(x.message)[:120] for x in w])
print(S.plan())
t0 = time.time(); losses = S.train(); torch.cuda.synchronize()
print('20 outer iterations at (48,16,100), config-1 size: %.2f s; losses %s ... %s' % (time.time() - t0, losses[:2], losses[-2:]))
import json; print('L2', json.load(open('L2_NODE_5.json')))
