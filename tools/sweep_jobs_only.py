"""a few launches of the duo parameter sweep with 1, 2 and 3 concurrent jobs of 4096 paths (for counter passes; the launches
are told apart by their grid size)"""
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
jobs = [mkjob(4096) for _ in range(3)]
for J in (1, 2, 3):
    for _ in range(4):
        KN.ode_bwd_multi(jobs[:J], t, th, *M, want_x=False, want_params=True)
torch.cuda.synchronize()
