"""Calibration of bench.py's cpu_baseline: time the oracle (oracle/refspec.py, the CPU port) on the headline workload in
THIS container -- the host BASELINE.md section 2 timed the reference itself on (8-core Xeon 2.1 GHz: 0.075 sub-steps/s at
8 threads and at 1 thread) -- and write the ratio to profiles/r02_oracle_calibration.json.
usage: python tools/calibrate_oracle.py        (CPU only, ~1 min)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from oracle import refspec as R
from bench import workload_params
params = workload_params(20, 4096, 4096, 32)
funcs = dict(h=P.func_h, f=P.func_f, g=P.func_g, a=P.func_a, b=P.func_b, c=P.func_c)
out = {'host': '8-core Intel Xeon 2.1 GHz (build container, same host as BASELINE.md section 2)',
       'workload': 'Ex4_1 cube d=20 N_r=N_b=4096 N_t=32, 1 generator + 1 discriminator sub-step',
       'reference_steps_per_s': 0.075, 'reference_source': 'BASELINE.md section 2 (train() of the reference, 8 threads = 1 thread)'}
for thr in (os.cpu_count(), 1):
    torch.set_num_threads(thr)
    torch.manual_seed(0)
    O = R.Solver(params, funcs, u_sol=P.func_u_sol, p=2)
    O.new_sample()
    t0 = time.perf_counter(); O.generator_step(); O.discriminator_step(); el = time.perf_counter() - t0
    key = 'oracle_steps_per_s_%d_threads' % thr
    out[key] = round(2 / el, 4)
    out['reference_over_oracle_%d_threads' % thr] = round(0.075 / (2 / el), 3)
    print(key, out[key])
out['reading'] = ('the port is faster than the reference on the same host (no d^2 Python coefficient loop for identity a, no torch.stack of '
                  'd^2 products): multiply cpu_baseline.value by reference_over_oracle to estimate the reference on the GPU box\'s host')
json.dump(out, open('profiles/r02_oracle_calibration.json', 'w'), indent=1)
