"""host seconds per phase of train()'s pipelined loop at the headline configuration (NODE_WAN_solver._phase_seconds):
   python tools/train_phases.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=5), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
os.makedirs('/tmp/tt', exist_ok=True); os.chdir('/tmp/tt')
if 'nooverlap' in sys.argv:     # the diagnostic and the next sample's refill one after the other on the main stream (as up to round 5)
    S.overlap_diagnostic = False
S.train()
n = S.iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if 'sync' in sys.argv:          # the synchronous loop (what a stop callback or report=True selects), with and without the captured refill
    S.pipeline = False
    for cap in (True, False, True, False):
        S.capture_refill = cap
        torch.cuda.synchronize(); t0 = time.perf_counter(); S.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('synchronous loop, capture_refill=%-5s: %.3f ms per outer iteration' % (cap, 1e3 * dt / n))
    sys.exit(0)
if 'stop' in sys.argv:          # what main.py of the reference runs: a stop hook (configs' acceptance rule) after every generator sub-iteration
    never = lambda solver, points, domain: P.stop(solver, points, domain) and False     # noqa: E731  (evaluated, never taken)
    for hook, name in ((None, 'no hook (pipelined loop)'), (never, "configs' stop rule evaluated after every generator sub-iteration")):
        S.stop = hook
        S.train()
        torch.cuda.synchronize(); t0 = time.perf_counter(); S.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('%-70s %.3f ms per outer iteration' % (name, 1e3 * dt / n))
    import cProfile, pstats, io
    S.iterations = 50
    pr = cProfile.Profile(); pr.enable(); S.train(); torch.cuda.synchronize(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(28); print(st.getvalue()[:7000])
    sys.exit(0)
if 'calls' in sys.argv:         # bench.py's pattern: train() in 25-iteration calls (the fixed cost of a call)
    S.iterations = 25
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        S.train()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('train() in calls of 25 outer iterations: %.3f ms per outer iteration' % (1e3 * dt / 200))
    sys.exit(0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); S.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%.3f ms per outer iteration; host ms per phase: %s' % (1e3 * dt / n, '  '.join('%s %.3f' % (k, 1e3 * v / n) for k, v in S._phase_seconds.items())))
