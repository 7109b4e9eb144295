"""wall time per outer iteration of train() at the headline configuration, pipelined (default) against the synchronous loop,
and a bitwise comparison of what the two leave behind (parameters, loss list, files)"""
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
os.makedirs('/tmp/tr', exist_ok=True)
res = {}
for pipe, devs in ((True, False), (False, False), (True, True), (False, True), (True, False), (False, False), (True, True), (False, True)):
    torch.manual_seed(0)
    S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=5), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f,
                        P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
    S.pipeline = pipe
    S.device_sampling = devs
    os.chdir('/tmp/tr')
    S.train()
    S.iterations = 200
    torch.cuda.synchronize(); t0 = time.perf_counter(); losses = S.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sig = hashlib.sha1(S.engine.theta.data.cpu().numpy().tobytes() + S.engine.phi.data.cpu().numpy().tobytes()).hexdigest()[:12]
    files = hashlib.sha1(open('losses_NODE_20.json', 'rb').read() + open('L2_NODE_20.json', 'rb').read()).hexdigest()[:12]
    best = torch.load('best_model_weights_NODE.pth')
    bsig = hashlib.sha1(b''.join(v.cpu().numpy().tobytes() for v in best.values())).hexdigest()[:12]
    print('device_sampling=%-5s pipeline=%-5s: %.2f ms per outer iteration -> %.0f sub-steps/s   params %s  files %s  best weights %s  n_losses %d' % (
        devs, pipe, 1e3 * dt / 200, 600 / dt, sig, files, bsig, len(losses)))
