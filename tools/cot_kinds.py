"""the duo sweep (and the narrow sweep) timed with the different kinds of cotangent: stored ubar, the initial-penalty residual
(first_only), the boundary residual (every time index), the weak form's dI/du formed in the sweep -- usage: python tools/cot_kinds.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L, d, H, K, m = 32, 20, 20, 10, 8
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
th = (0.3 * torch.randn(_lib.lib.xw_theta_size(d, H, K), generator=g, dtype=torch.float64)).to(dev)
r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dev)
xT, t, start = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev), torch.linspace(0, 1, L, dtype=torch.float64).to(dev), r(N)
u, Y = torch.empty(L, N, dtype=torch.float64, device=dev), torch.empty(L, H, N, dtype=torch.float64, device=dev)
act = torch.empty(L - 1, KN.ode_act_rows(1, H, K, m), KN.ode_act_cols(N), dtype=torch.float64, device=dev)
slab = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=torch.float64, device=dev)
M = (1, H, K, m)
job = dict(xT=xT, start=start, u=u, Y=Y, act=act)
KN.ode_fwd_multi([job], t, th, *M)
ubar, v, w, href, gref = r(L, N), r(L, N), torch.rand(N, generator=g, dtype=torch.float64).to(dev), r(N), r(L, N)
kinds = {'stored ubar': dict(ubar=ubar), 'ones': dict(ubar=None),
         'residual, first index only (sweep A)': dict(res=dict(u=u, ref=href, coef=0.3, base=1.0, first_only=True)),
         'residual, every index (boundary sweep)': dict(res=dict(u=u, ref=gref, coef=0.3, base=0.0, first_only=False)),
         'weak form dI/du (sweep B)': dict(res=dict(u=u, ref=v, coef=0.1, base=0.2, weak=dict(w=w, ckappa=-1.0)))}
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, kw in kinds.items():
    a = timeit(lambda: KN.ode_bwd_multi([dict(job, gslab=slab, **kw)], t, th, *M, want_x=False, want_params=True))
    b = timeit(lambda: KN.ode_bwd_multi([dict(job, gslab=slab, **kw)], t, th, *M, want_x=False, want_params=True, narrow=True))
    print('%-42s duo %6.1f us   narrow %6.1f us' % (name, a, b))
