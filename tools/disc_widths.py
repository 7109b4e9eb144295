import os, sys
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, q = 4096, 32, 20, 9
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
for W in (50, 64, 96, 128):
    ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
    xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
    v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
    gx = torch.empty(d, N, dtype=torch.float64, device=dev); gt = torch.empty(N, dtype=torch.float64, device=dev)
    vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
    slv = torch.empty(KN.disc_bwd_slabs(N, L), ph.numel(), dtype=torch.float64, device=dev)
    rec = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(L * N), dtype=torch.float64, device=dev)
    def tm(fn, n=10):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n
    a = tm(lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gx, gtv=gt, ngrad=N))
    b = tm(lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gx, gtv=gt, ngrad=N, act=rec))
    c = tm(lambda: KN.disc_bwd(xT, t, ph, vbar, W, q, gslab=slv, act=rec))
    print('W=%d: fwd %.1f us, fwd+record %.1f us, bwd from record %.1f us' % (W, a, b, c))
