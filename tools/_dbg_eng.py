import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import configs.Ex4_1_funcs as P
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
z = np.load('tests/golden/ref_d20_small_midpoint.npz'); params = json.loads(str(z['params_json']))
torch.manual_seed(int(z['seed'])); np.random.seed(int(z['seed']))
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
s = S.setup
domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
pts = Comb_loader(s['N_r'], s['N_b'], domain, S.device)
eng = S.engine
G = eng.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
eng.generator_step(G)
torch.cuda.synchronize()
out = dict(u=G.u.cpu(), ub=G.ub.cpu(), act=G.act.cpu(), act_b=G.act_b.cpu(), slabA=G.slabA.sum(0).cpu(), slabB=G.slabB.sum(0).cpu(), grad=eng.grad_u.cpu(),
           gx=G.gx.cpu(), gs=G.gs.cpu(), v=G.v.cpu(), scal=eng.scal.cpu(), slabA_i=G.slabA[:G.ns_u].sum(0).cpu(), slabA_b=G.slabA[G.ns_u:].sum(0).cpu())
torch.save(out, sys.argv[1])
