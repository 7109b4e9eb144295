"""a stepper forward ALONE on the chip (the diagnostic's, the stop hook's, Engine.predict): wide against narrow tiles, no stores but u --
microseconds per launch:  python tools/lone_forward.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
dev = torch.device('cuda')
H, K, m = 20, 10, 8


def timed(f, n=40):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N, L, d in ((512, 32, 20), (2048, 32, 20), (4096, 32, 20), (8192, 32, 20), (16384, 32, 20), (4096, 64, 50), (16384, 64, 50)):
    g = torch.Generator().manual_seed(0)
    th = (0.2 * torch.randn(_lib.lib.xw_theta_size(d, H, K), generator=g, dtype=torch.float64)).to(dev)
    xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
    s = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
    out = []
    for method in (1,):
        us = {}
        for narrow in (False, True):
            u = torch.empty(L, N, dtype=torch.float64, device=dev)
            job = [dict(xT=xT, start=s, u=u, Y=None)]
            us[narrow] = (timed(lambda: KN.ode_fwd_multi(job, t, th, method, H, K, m, narrow=narrow)), u.clone())
        out.append('wide %7.1f us  narrow %7.1f us  max|du| %.1e' % (us[False][0], us[True][0], float((us[False][1] - us[True][1]).abs().max())))
    print('N %6d L %3d d %3d : %s' % (N, L, d, out[0]), flush=True)
