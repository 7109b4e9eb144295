"""a few launches of the test-network kernels at the headline size (for counter passes)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, W, q = 4096, 32, 20, int(os.environ.get('XW_W', '50')), 9     # (XW_W=128: the wide container)
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
slv = torch.empty(KN.disc_bwd_slabs(N, L), ph.numel(), dtype=torch.float64, device=dev)
rec = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(L * N), dtype=torch.float64, device=dev) if os.environ.get('XW_RECORD', '1') == '1' else None
for _ in range(5):
    KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N, act=rec)
    KN.disc_bwd(xT, t, ph, vbar, W, q, gslab=slv, act=rec)
torch.cuda.synchronize()
