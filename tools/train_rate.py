"""wall time per outer iteration of train() at the headline configuration: reference-parity host sampling (default) against
device sampling (solver.device_sampling = True: no seed parity with the reference).  (Handing the JSON / weight files to a writer thread was
measured too: 4.0-4.9 ms against 4.1-4.2 ms -- pickle holds the interpreter lock, nothing overlaps; not kept.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
os.makedirs('/tmp/tr', exist_ok=True)
for rep in range(2):
    for dev_sampling in (False, True):
        for _ in (0,):
            torch.manual_seed(0)
            S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=5), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f,
                                P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
            S.device_sampling = dev_sampling
            os.chdir('/tmp/tr')
            S.train()
            S.iterations = 100
            torch.cuda.synchronize(); t0 = time.perf_counter(); S.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print('device_sampling=%-5s: %.2f ms per outer iteration (3 sub-steps + resample + diagnostic + files) -> %.0f sub-steps/s' % (
                dev_sampling, 1e3 * dt / 100, 300 / dt))
