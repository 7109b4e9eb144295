"""wall time of the sections of one outer iteration of train() at the headline configuration (synchronised after each)"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
torch.set_num_threads(int(os.environ.get("XW_HOST_THREADS", torch.get_num_threads()))); print("host threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)), os.cpu_count())
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=3), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f,
                    P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/pt', exist_ok=True); os.chdir('/tmp/pt')
S.train()
eng = S.engine
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
n = 20
for k in range(n):
    t = time.perf_counter()
    domain = S._new_domain(); t = tick('new_domain', t)
    points = S._loader(domain); t = tick('loader (host sampling)', t)
    L2 = S._l_norm(points, domain.V()); t = tick('L_norm #1', t)
    shards = S._shard(S._groups(points)); t = tick('groups/shard', t)
    groups = [eng.load_group(du, dv, bd, domain, ng, nbg, into=old) for (du, dv, bd, ng, nbg), old in zip(shards, S._group_cache)]
    S._group_cache = groups; t = tick('load_group (upload + tabulate)', t)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); eng.generator_step(groups[0]); ev[1].record(); x = eng.loss_u().item()
    eng.generator_step(groups[0]); ev[2].record(); x = eng.loss_u().item()
    acc['  (gen step 1, GPU events)'] = acc.get('  (gen step 1, GPU events)', 0.0) + ev[0].elapsed_time(ev[1]) * 1e-3
    acc['  (gen step 2, GPU events)'] = acc.get('  (gen step 2, GPU events)', 0.0) + ev[1].elapsed_time(ev[2]) * 1e-3
    t = tick('2 generator steps + .item()', t)
    json.dump([x], open('losses.json', 'w')); torch.save(S.u_net.state_dict(), 'w.pth'); t = tick('json + torch.save', t)
    eng.discriminator_step(groups[0]); x = eng.loss_v().item(); t = tick('discriminator step + .item()', t)
    points = S._loader(domain); t = tick('loader #2', t)
    L2 = S._l_norm(points, domain.V()); t = tick('L_norm #2', t)
for k_, v in acc.items():
    print('%-36s %7.2f ms' % (k_, 1e3 * v / n))
print('%-36s %7.2f ms' % ('total', 1e3 * sum(acc.values()) / n))
