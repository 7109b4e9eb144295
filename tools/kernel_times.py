"""time the stepper kernels alone at the headline size (N=4096 paths, L=32, d=20, midpoint)
usage: python tools/kernel_times.py [method]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
method = sys.argv[1] if len(sys.argv) > 1 else 'midpoint'
N, L, d, H, K, m = int(os.environ.get("XW_N", 4096)), 32, 20, 20, 10, 8
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
P = _lib.lib.xw_theta_size(d, H, K, m)
th = (0.3 * torch.randn(P, generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
start = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
ubar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
u = torch.empty(L, N, dtype=torch.float64, device=dev); Y = torch.empty(L, H, N, dtype=torch.float64, device=dev)
gx = torch.empty(d, N, dtype=torch.float64, device=dev); gs = torch.empty(N, dtype=torch.float64, device=dev)
slab = torch.empty(KN.ode_bwd_slabs(N), P, dtype=torch.float64, device=dev)
job = dict(xT=xT, start=start, u=u, Y=Y)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = (KN.METHODS[method], H, K, m)
print('fwd      %.1f us' % timeit(lambda: KN.ode_fwd_multi([job], t, th, *M)))
print('bwd x    %.1f us' % timeit(lambda: KN.ode_bwd_multi([dict(job, ubar=None, gx=gx, gs=gs)], t, th, *M, want_x=True, want_params=False)))
print('bwd par  %.1f us' % timeit(lambda: KN.ode_bwd_multi([dict(job, ubar=ubar, gslab=slab)], t, th, *M, want_x=False, want_params=True)))
print('checksum %.12e %.12e' % (float(slab.sum(0).abs().sum()), float(gx.abs().sum())))
ar = KN.ode_act_rows(M[0], H, K, m)
if ar:
    act = torch.empty(L - 1, ar, N, dtype=torch.float64, device=dev)
    joba = dict(job, act=act)
    print('fwd+act  %.1f us' % timeit(lambda: KN.ode_fwd_multi([joba], t, th, *M)))
    print('bwd x  (act) %.1f us' % timeit(lambda: KN.ode_bwd_multi([dict(joba, ubar=None, gx=gx, gs=gs)], t, th, *M, want_x=True, want_params=False)))
    print('bwd par(act) %.1f us' % timeit(lambda: KN.ode_bwd_multi([dict(joba, ubar=ubar, gslab=slab)], t, th, *M, want_x=False, want_params=True)))
    print('checksum %.12e %.12e' % (float(slab.sum(0).abs().sum()), float(gx.abs().sum())))
# ---- test network at the headline size (131072 points)
W, q = 50, 9
Pv = _lib.lib.xw_phi_size(d, W)
ph = (0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
slv = torch.empty(KN.disc_bwd_slabs(N, L), Pv, dtype=torch.float64, device=dev)
print('disc fwd %.1f us' % timeit(lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N)))
print('disc bwd %.1f us' % timeit(lambda: KN.disc_bwd(xT, t, ph, vbar, W, q, gslab=slv)))
print('checksum %.12e %.12e %.12e' % (float(v.abs().sum()), float(vt.abs().sum()), float(slv.sum(0).abs().sum())))
