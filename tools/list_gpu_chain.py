"""Ball domains at config-5 size: how long does the GPU need for the sub-steps of ONE outer iteration (n1 = 2 generator + n2 = 1
discriminator sub-iterations over all groups of a sample), and how long does the host need to queue them?  The first is the floor
of an outer iteration whatever the host does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import configs.Ex4_3_funcs as P
from src.training import NODE_WAN_solver
name = sys.argv[1] if len(sys.argv) > 1 else 'NSphere_THourglass'
params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 10, 'N_t': 20, 'N_r': 8192, 'N_b': 8192, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 2, 'domain': name}
torch.manual_seed(0); np.random.seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/c5', exist_ok=True); os.chdir('/tmp/c5')
S.train()
eng, groups = S.engine, S._group_cache
for G in groups:
    G.persistent = False
def cycle():
    for _ in range(2):
        eng.begin_substep('u', True)
        for G in groups:
            eng.generator_step(G)
    eng.begin_substep('v', True)
    for G in groups:
        eng.discriminator_step(G)
host, total = [], []
for _ in range(12):
    torch.cuda.synchronize()
    t = time.perf_counter(); cycle(); h = time.perf_counter() - t; torch.cuda.synchronize(); total.append(time.perf_counter() - t); host.append(h)
host.sort(); total.sort()
print('%s, %d groups (%s paths): the sub-steps of one outer iteration -- host queues them in %.2f ms, the GPU has finished them after %.2f ms (medians of 12)'
      % (name, len(groups), ' '.join(str(G.N) for G in groups), 1e3 * host[6], 1e3 * total[6]))
