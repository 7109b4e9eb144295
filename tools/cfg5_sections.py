"""wall time of the sections of one outer iteration of BASELINE config 5 (d=10 cone, N_r=N_b=8192, N_t=20), synchronised
after each section (so the sum exceeds train()'s iteration time, where host and GPU work overlap)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import configs.Ex4_3_funcs as P
from src.training import NODE_WAN_solver
name = sys.argv[1] if len(sys.argv) > 1 else 'NSphere_TCone'
params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 10, 'N_t': 20, 'N_r': 8192, 'N_b': 8192, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 2, 'domain': name}
torch.manual_seed(0); np.random.seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './',
                    func_u_sol=getattr(P, 'func_u_sol', None), p=2)
os.makedirs('/tmp/c5', exist_ok=True); os.chdir('/tmp/c5')
S.train()
if S.host_threads:
    torch.set_num_threads(min(torch.get_num_threads(), int(S.host_threads)))
print('host threads', torch.get_num_threads())
eng = S.engine
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
n = 10
for k in range(n):
    t = time.perf_counter()
    domain = S._new_domain(); t = tick('new_domain', t)
    points = S._loader(domain); t = tick('loader (host sampling)', t)
    shards = S._shard(S._groups(points)); t = tick('groups/shard', t)
    if len(S._group_cache) != len(shards):
        S._group_cache = [None] * len(shards)
    tabs = eng.tabulate_sample([sh[:3] for sh in shards], domain); t = tick('tabulate_sample', t)
    big = max(range(len(shards)), key=lambda i: shards[i][0].shape[0] * shards[i][0].shape[1])
    groups = [eng.load_group(du, dv, bd, domain, ng, nbg, into=old, shared_grid_t0=S._grid_hint, tab=tb, verify=(i == big))
              for i, ((du, dv, bd, ng, nbg), old, tb) in enumerate(zip(shards, S._group_cache, tabs))]
    S._group_cache = groups; t = tick('load_group x%d' % len(groups), t)
    for G in groups:
        G.persistent = False
    for _ in range(2):
        eng.begin_substep('u', True)
        t1 = time.perf_counter()
        lu = []
        for G in groups:
            eng.generator_step(G); lu.append(eng.loss_u().clone())
        acc['  generator host time'] = acc.get('  generator host time', 0.0) + time.perf_counter() - t1
        lu = torch.stack(lu).tolist()
        t = tick('generator sub-iteration (all groups)', t)
    eng.begin_substep('v', True)
    t1 = time.perf_counter()
    for G in groups:
        eng.discriminator_step(G)
    acc['  discriminator host time'] = acc.get('  discriminator host time', 0.0) + time.perf_counter() - t1
    x = eng.loss_v().item(); t = tick('discriminator sub-iteration (all groups)', t)
    points = S._loader(domain); t = tick('loader #2', t)
    L2 = S._l_norm(points, domain.V()); t = tick('L_norm', t)
for k_, v in acc.items():
    print('%-44s %7.2f ms' % (k_, 1e3 * v / n))
print('groups', len(groups), [int(G.N) for G in groups], [int(G.L) for G in groups])
