"""k_disc_fwd alone, plain against the hoisted x-projection (xw_disc_xproj + xw_disc_fwd_xproj), at the shapes of BASELINE.json's
configurations: microseconds per launch (HIP events over 50 launches), generator form (no record) and discriminator form (record)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib

dev = torch.device('cuda')
W, q = 50, 9


def timed(f, n=50):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N, L, d in ((4096, 32, 20), (16384, 64, 50), (65536 // 8, 128, 100), (4096, 32, 5), (4096, 32, 50), (4096, 32, 100)):
    g = torch.Generator().manual_seed(0)
    ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
    xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
    v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
    gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
    rec = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(L * N), dtype=torch.float64, device=dev)
    xp = torch.empty(64, N, dtype=torch.float64, device=dev)
    row = []
    for act in (None, rec):
        plain = timed(lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N, act=act))
        v0 = v.clone()
        proj = timed(lambda: KN.disc_xproj(xT, ph, W, out=xp))
        hoist = timed(lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N, act=act, xproj=xp))
        err = float((v - v0).abs().max())
        row.append('%s plain %8.1f us   hoisted %8.1f us (+ table %5.1f us)   %+5.1f %%   max|dv| %.1e'
                   % ('record' if act is not None else 'no rec', plain, hoist, proj, 100 * (hoist + proj - plain) / plain, err))
    print('N %6d L %4d d %4d :  %s\n%27s%s' % (N, L, d, row[0], '', row[1]), flush=True)
