import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=3), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/pt', exist_ok=True); os.chdir('/tmp/pt')
S.train()
torch.set_num_threads(4)
eng = S.engine
pr = cProfile.Profile()
tot = 0.0
for k in range(20):
    domain = S._new_domain(); points = S._loader(domain)
    shards = S._shard(S._groups(points))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable()
    groups = [eng.load_group(du, dv, bd, domain, ng, nbg, into=old, shared_grid_t0=S._grid_hint) for (du, dv, bd, ng, nbg), old in zip(shards, S._group_cache)]
    pr.disable()
    torch.cuda.synchronize(); tot += time.perf_counter() - t0
    S._group_cache = groups
print('load_group %.2f ms per call (synchronised)' % (1e3 * tot / 20))
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(22); print(st.getvalue()[:5000])
