import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from utils.auxillary_funcs import rel_err
torch.manual_seed(0)
prm = dict(workload_params(5, 1024, 1024, 16), iterations=60, adjoint=True)
S = NODE_WAN_solver(prm, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/adj', exist_ok=True); os.chdir('/tmp/adj')
t0 = time.time(); S.train(); torch.cuda.synchronize()
print('adjoint=True: 60 outer iterations in %.2f s, graphs %s, loss_u %.4g, loss_v %.4g' % (time.time() - t0, S.engine.use_graphs, S.last_loss_u, S.last_loss_v))
dom = S.domain([-1, 1], 5, 0, 1, 16); X = dom.interior(2048)
print('rel L2', float(rel_err(X, S.u_net, P.func_u_sol, 2, dom.V(), 2048)))
