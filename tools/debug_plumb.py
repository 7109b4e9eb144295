import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import configs.Ex4_1_funcs as P
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
from utils.auxillary_funcs import L_norm
z = np.load('tests/golden/ref_plumb_midpoint.npz'); params = json.loads(str(z['params_json']))
torch.manual_seed(int(z['seed']))
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
s = S.setup
domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
pts = Comb_loader(s['N_r'], s['N_b'], domain, S.device)
X = pts.interioru
with torch.no_grad():
    u = S.u_net(X).squeeze(2).cpu()
ref = torch.from_numpy(z['gen1/u'])
print('max |u - ref|', float((u - ref).abs().max()), 'at', int((u-ref).abs().argmax()))
t = P.func_u_sol(X).detach()
for name, dev in (('cpu', 'cpu'), ('gpu', 'cuda')):
    diff = (t.to(dev) - u.to(dev))
    m = torch.mean(torch.abs(diff) ** 2)
    print(name, 'mean', repr(float(m)), 'norm', repr(float((32.0 * m) ** 0.5)), 'sumsq', repr(float((diff*diff).sum()/diff.numel())))
print('golden', repr(float(z['L2_start'])))
print('L_norm', repr(float(L_norm(X, S.u_net, 2, P.func_u_sol, domain.V(), 256))))
