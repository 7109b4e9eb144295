import os, sys
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, H, K, m = int(os.environ.get('DBG_N', 96)), 12, int(os.environ.get('DBG_D', 20)), 20, 10, 8
g = torch.Generator().manual_seed(1)
P = _lib.lib.xw_theta_size(d, H, K)
th = (0.3 * torch.randn(P, generator=g, dtype=torch.float64)).cuda()
xT = torch.rand(d, N, generator=g, dtype=torch.float64).cuda()
t = torch.linspace(0, 1, L, dtype=torch.float64).cuda()
start = torch.randn(N, generator=g, dtype=torch.float64).cuda()
mid = 1
rows = KN.ode_act_rows(mid, H, K, m)
out = {}
for xo in (False, True):
    u = torch.empty(L, N, dtype=torch.float64, device='cuda'); Y = torch.empty(L, H, N, dtype=torch.float64, device='cuda')
    act = torch.full((L - 1, rows, KN.ode_act_cols(N)), float('nan'), dtype=torch.float64, device='cuda')
    KN.ode_fwd_multi([dict(xT=xT, start=start, u=u, Y=Y, act=act)], t, th, mid, H, K, m, act_x_only=xo)
    torch.cuda.synchronize()
    out['act%d' % xo], out['u%d' % xo] = act.cpu(), u.cpu()
torch.save(out, sys.argv[1])
