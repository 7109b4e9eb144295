import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, W, q = 4096, 32, 20, 50, 9
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
gx = torch.empty(d, N, dtype=torch.float64, device=dev); gt = torch.empty(N, dtype=torch.float64, device=dev)
rec = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(L * N), dtype=torch.float64, device=dev)
for cap in (256, 320, 384, 448, 512):
    out = []
    for act in (None, rec):
        for _ in range(3):
            KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gx, gtv=gt, ngrad=N, max_blocks=cap, act=act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gx, gtv=gt, ngrad=N, max_blocks=cap, act=act)
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / 20)
    print('dyn=%s cap %d: fwd+grad %.1f us   +record %.1f us' % (os.environ.get('XW_DISC_DYNAMIC', '1'), cap, out[0], out[1]))
