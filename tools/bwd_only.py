"""time the test-network backward (from the record, and recomputing) at the headline size"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, W, q = 4096, 32, int(os.environ.get('D', 20)), 50, 9
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
slv = torch.empty(KN.disc_bwd_slabs(N, L), ph.numel(), dtype=torch.float64, device=dev)
rec = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(L * N), dtype=torch.float64, device=dev)
KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, act=rec)
for use in (rec, None):
    for _ in range(3):
        KN.disc_bwd(xT, t, ph, vbar, W, q, gslab=slv, act=use)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        KN.disc_bwd(xT, t, ph, vbar, W, q, gslab=slv, act=use)
    e1.record(); torch.cuda.synchronize()
    print('blocks %d  %s: %.1f us' % (slv.shape[0], 'record' if use is not None else 'recompute', 1e3 * e0.elapsed_time(e1) / 20))
