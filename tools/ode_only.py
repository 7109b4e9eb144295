"""a few launches of the stepper kernels at the headline size (for counter passes)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, H, K, m = 4096, 32, 20, 20, 10, 8
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
th = (0.3 * torch.randn(_lib.lib.xw_theta_size(d, H, K), generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
start = torch.randn(N, generator=g, dtype=torch.float64).to(dev); ubar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
u = torch.empty(L, N, dtype=torch.float64, device=dev); Y = torch.empty(L, H, N, dtype=torch.float64, device=dev)
gx = torch.empty(d, N, dtype=torch.float64, device=dev); gs = torch.empty(N, dtype=torch.float64, device=dev)
slab = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=torch.float64, device=dev)
M = (1, H, K, m)
act = torch.empty(L - 1, KN.ode_act_rows(1, H, K, m), N, dtype=torch.float64, device=dev)
job = dict(xT=xT, start=start, u=u, Y=Y, act=act)
for _ in range(5):
    KN.ode_fwd_multi([job], t, th, *M)
    KN.ode_bwd_multi([dict(job, ubar=None, gx=gx, gs=gs)], t, th, *M, want_x=True, want_params=False)
    KN.ode_bwd_multi([dict(job, ubar=ubar, gslab=slab)], t, th, *M, want_x=False, want_params=True)
    # the same three on narrow tiles (csrc/xw_ode_n4.h: k_ode_fwd_n4 / k_ode_bwd_n4)
    KN.ode_fwd_multi([job], t, th, *M, narrow=True)
    KN.ode_bwd_multi([dict(job, ubar=None, gx=gx, gs=gs)], t, th, *M, want_x=True, want_params=False, narrow=True)
    KN.ode_bwd_multi([dict(job, ubar=ubar, gslab=slab)], t, th, *M, want_x=False, want_params=True, narrow=True)
torch.cuda.synchronize()
