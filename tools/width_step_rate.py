"""sub-steps/s of the g,g,d cycle on the headline sample (d = 20, 4096 + 4096 paths, N_t = 32, midpoint, 8 / 9 layers) at other
network widths -- what the wide containers of round 6 buy a configuration that used to run on the generic path:
    python tools/width_step_rate.py            -> profiles/r06_width_step_rate.txt"""
import os
import sys
import time
import warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
import configs.Ex4_1_funcs as P
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader

dev = torch.device('cuda', 0)
cases = [(20, 10, 50, 8), (32, 12, 64, 8), (48, 16, 100, 8), (64, 16, 128, 8), (64, 16, 128, 10), (20, 10, 128, 8), (20, 10, 96, 8), (64, 16, 50, 8)]
if len(sys.argv) > 1 and sys.argv[1] == 'generic':
    cases = [(20, 10, 50, 12)]          # u_layers > 10: what is left on the generic stepper
for (H, K, W, m) in cases:
    params = B.workload_params(20, 4096, 4096, 32)
    params.update(u_hidden_dim=H, u_hidden_hidden_dim=K, v_hidden_dim=W, u_layers=m)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, dev, './', func_u_sol=P.func_u_sol, p=2)
    eng, s = S.engine, S.setup
    domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
    torch.manual_seed(1000)
    du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, dev)[0]
    G = eng.load_group(du, dv, bd, domain)

    def cycle(n):
        for _ in range(n):
            eng.generator_step(G)
            eng.generator_step(G)
            eng.discriminator_step(G)
    cycle(2 if eng.generic[0] else 6)
    torch.cuda.synchronize()
    n = 2 if eng.generic[0] else 40
    t0 = time.perf_counter()
    cycle(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('u_theta (%d, %d) x %d layers in %s%s, v_phi %d in %d%s: %8.1f sub-steps/s  (%.3f ms per sub-step; loss_v %.6e)'
          % (H, K, m, (eng.H, eng.K), ' GENERIC' if eng.generic[0] else '', W, eng.W, ' GENERIC' if eng.generic[1] else '',
             3 * n / dt, 1e3 * dt / (3 * n), float(eng.scal[5])), flush=True)
    del S, eng, G
