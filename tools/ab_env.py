"""engine switches read from the environment at construction, compared in ONE process on the headline workload (boxes of the pool
differ by several per cent; only same-process numbers compare):   python tools/ab_env.py XW_FUSE_DISC_COT=0,1 [rounds]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
name, vals = sys.argv[1].split('=')
vals = vals.split(',')
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda')
n = 40


def run():
    torch.manual_seed(0)
    S = NODE_WAN_solver(workload_params(20, 4096, 4096, 32), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, dev, './',
                        func_u_sol=P.func_u_sol, p=2)
    eng, s = S.engine, S.setup
    domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
    du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, dev)[0]
    G = eng.load_group(du, dv, bd, domain)
    out = {}
    for key, fn in (('gen', eng.generator_step), ('disc', eng.discriminator_step)):
        for _ in range(6):
            fn(G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn(G)
        torch.cuda.synchronize(); out[key] = 1e3 * (time.perf_counter() - t0) / n
    for _ in range(4):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize()
    out['cycle'] = 1e3 * (time.perf_counter() - t0) / n
    return out, float(eng.theta.data.double().sum()), float(eng.phi.data.double().sum())


for r in range(rounds):
    for v in vals:
        os.environ[name] = v
        o, cu, cv = run()
        print('%s=%-6s generator %.4f ms  discriminator %.4f ms  cycle %.4f ms -> %.0f sub-steps/s   sums %.12e %.12e' % (
            name, v, o['gen'], o['disc'], o['cycle'], 3e3 / o['cycle'], cu, cv), flush=True)
