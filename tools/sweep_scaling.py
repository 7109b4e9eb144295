"""param sweep (duo kernel) time against the number of paths / concurrent jobs: is it bound by SIMD sharing or by the memory
system?  One job of N paths; J jobs of 4096 paths; and the same J-job launch with all jobs reading ONE activation store."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
L, d, H, K, m = 32, 20, 20, 10, 8
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
th = (0.3 * torch.randn(_lib.lib.xw_theta_size(d, H, K), generator=g, dtype=torch.float64)).to(dev)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
M = (1, H, K, m)
def mkjob(N):
    xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
    start = torch.randn(N, generator=g, dtype=torch.float64).to(dev); ubar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
    u = torch.empty(L, N, dtype=torch.float64, device=dev); Y = torch.empty(L, H, N, dtype=torch.float64, device=dev)
    act = torch.empty(L - 1, KN.ode_act_rows(1, H, K, m), KN.ode_act_cols(N), dtype=torch.float64, device=dev)
    slab = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=torch.float64, device=dev)
    job = dict(xT=xT, start=start, u=u, Y=Y, act=act)
    KN.ode_fwd_multi([job], t, th, *M)
    return dict(job, ubar=ubar, gslab=slab)
def timeit(jobs, n=20):
    for _ in range(3):
        KN.ode_bwd_multi(jobs, t, th, *M, want_x=False, want_params=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        KN.ode_bwd_multi(jobs, t, th, *M, want_x=False, want_params=True)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for N in (256, 1024, 2048, 4096, 4112, 8192, 12288, 16384):
    print('1 job of %5d paths: %.1f us' % (N, timeit([mkjob(N)])))
jobs = [mkjob(4096) for _ in range(3)]
for J in (1, 2, 3):
    print('%d jobs of 4096 paths, own stores: %.1f us' % (J, timeit(jobs[:J])))
same = [dict(jobs[0], gslab=j['gslab']) for j in jobs]
for J in (2, 3):
    print('%d jobs of 4096 paths, ONE store (same reads): %.1f us' % (J, timeit(same[:J])))
