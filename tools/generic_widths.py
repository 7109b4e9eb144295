"""kernel times of the generic-width path (csrc/xw_generic.hip) next to the MFMA instantiations, at the headline sample size:
   python tools/generic_widths.py [N L d]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from xnode_wan_pde_solver_amd import kernels as KN
N, L, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 32, 20)
dev, F64 = torch.device('cuda'), torch.float64
g = torch.Generator().manual_seed(0)
x = torch.rand(d, N, generator=g, dtype=F64).to(dev) * 2 - 1
t = torch.linspace(0, 1, L, dtype=F64).to(dev)
start = torch.randn(N, generator=g, dtype=F64).to(dev)
ubar = torch.randn(L, N, generator=g, dtype=F64).to(dev)


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print('N = %d paths, L = %d times, d = %d (%d points)' % (N, L, d, N * L))
for (H, K, m) in ((20, 10, 8), (32, 12, 8), (48, 16, 8), (64, 16, 8)):
    th = (0.2 * torch.randn(KN.theta_size(d, H, K), generator=g, dtype=F64)).to(dev)
    gen = KN.ode_generic(H, K)
    n = 3 if gen else 20
    u, Y = KN.ode_fwd(x, t, start, th, 1, H, K, m)
    tf = timeit(lambda: KN.ode_fwd(x, t, start, th, 1, H, K, m, u=u, Y=Y), n)
    gx, gs = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
    slab = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=F64, device=dev)
    tx = timeit(lambda: KN.ode_bwd(x, t, start, th, Y, ubar, 1, H, K, m, want_x=True, want_params=False, gx=gx, gs=gs), n)
    tp = timeit(lambda: KN.ode_bwd(x, t, start, th, Y, ubar, 1, H, K, m, want_x=True, want_params=True, gx=gx, gs=gs, gslab=slab), n)
    print('stepper (%2d,%2d) m=%d  %-8s forward %9.1f us   x-only sweep %9.1f us   sweep with weight gradients %10.1f us  (midpoint, no activation store)'
          % (H, K, m, 'generic' if gen else 'MFMA', tf, tx, tp))
for W in (50, 64, 100, 128):
    q = 9
    ph = (0.2 * torch.randn(KN.phi_size(d, W), generator=g, dtype=F64)).to(dev)
    gen = KN.disc_generic(W)
    n = 3 if gen else 20
    rec = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(N * L), dtype=F64, device=dev)
    v, vt = torch.empty(L, N, dtype=F64, device=dev), torch.empty(L, N, dtype=F64, device=dev)
    gxv, gtv = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
    tf = timeit(lambda: KN.disc_fwd(x, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N, act=rec), n)
    tb = timeit(lambda: KN.disc_bwd(x, t, ph, ubar, W, q, act=rec), n)
    print('test network W=%3d q=%d %-8s forward + record %10.1f us   reverse from the record %11.1f us' % (W, q, 'generic' if gen else 'MFMA', tf, tb))
