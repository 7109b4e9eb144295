"""block caps of the test network (Engine.v_blocks in the generator sub-steps, v_blocks_disc in the discriminator sub-step) swept
in ONE process on the headline workload:  python tools/sweep_caps.py [gen caps, comma separated] [disc caps]   (slots: 512)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
gen = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '288,320,352').split(',')]
disc = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '352,384,416,448,480').split(',')]
dev = torch.device('cuda')
n = 40


def run(vb, vbd):
    os.environ['XW_V_BLOCKS'], os.environ['XW_V_BLOCKS_DISC'] = str(vb), str(vbd)
    torch.manual_seed(0)
    S = NODE_WAN_solver(workload_params(20, 4096, 4096, 32), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, dev, './',
                        func_u_sol=P.func_u_sol, p=2)
    eng, s = S.engine, S.setup
    domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
    du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, dev)[0]
    G = eng.load_group(du, dv, bd, domain)
    out = {}
    for name, fn in (('gen', eng.generator_step), ('disc', eng.discriminator_step)):
        for _ in range(6):
            fn(G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn(G)
        torch.cuda.synchronize(); out[name] = 1e3 * (time.perf_counter() - t0) / n
    for _ in range(4):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
    torch.cuda.synchronize()
    out['cycle'] = 1e3 * (time.perf_counter() - t0) / n
    return out


for rep in range(2):
    for vb in gen:
        o = run(vb, disc[len(disc) // 2])
        print('gen cap %3d: generator sub-step %.4f ms   (cycle %.4f -> %.0f /s)' % (vb, o['gen'], o['cycle'], 3e3 / o['cycle']), flush=True)
    for vbd in disc:
        o = run(gen[len(gen) // 2], vbd)
        print('disc cap %3d: discriminator sub-step %.4f ms   (cycle %.4f -> %.0f /s)' % (vbd, o['disc'], o['cycle'], 3e3 / o['cycle']), flush=True)
