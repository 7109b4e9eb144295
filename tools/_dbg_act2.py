import os, sys
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
N, L, d, H, K, m = 37, 4, 20, 20, 10, 8
g = torch.Generator().manual_seed(1)
P = _lib.lib.xw_theta_size(d, H, K)
th = (0.3 * torch.randn(P, generator=g, dtype=torch.float64)).cuda()
xT = torch.rand(d, N, generator=g, dtype=torch.float64).cuda()
t = torch.linspace(0, 1, L, dtype=torch.float64).cuda()
start = torch.randn(N, generator=g, dtype=torch.float64).cuda()
for mid in (0, 1):
    rows = KN.ode_act_rows(mid, H, K, m)
    u = torch.empty(L, N, dtype=torch.float64, device='cuda'); Y = torch.empty(L, H, N, dtype=torch.float64, device='cuda')
    act = torch.full((L - 1, rows, KN.ode_act_cols(N)), float('nan'), dtype=torch.float64, device='cuda')
    KN.ode_fwd_multi([dict(xT=xT, start=start, u=u, Y=Y, act=act)], t, th, mid, H, K, m)
    torch.cuda.synchronize()
    A = act.reshape(L - 1, act.shape[2] // 16, rows, 16).cpu()
    bad = torch.nonzero(torch.isnan(A))
    print('method', mid, 'rows', rows, 'nan count', len(bad), 'first', bad[:6].tolist(), 'rows with nan', sorted(set(bad[:, 2].tolist())), 'tiles', sorted(set(bad[:, 1].tolist())))
