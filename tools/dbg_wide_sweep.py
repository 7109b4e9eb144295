import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import test_gpu_kernels as T
from oracle import refspec as R
from xnode_wan_pde_solver_amd import kernels as KN
Hh, Kk, m, solver = 32, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 3, sys.argv[2] if len(sys.argv) > 2 else 'rk4'
N, L, d = 37, 6, 5
cfg = dict(T._cfg(m, solver), u_hidden_dim=Hh, u_hidden_hidden_dim=Kk)
torch.manual_seed(31)
theta, _ = R.init_parameters(cfg, T._setup(d, 2))
for p in theta.values():
    if p.dim() == 1:
        p.copy_(0.3 * torch.randn(p.shape, dtype=torch.float64))
th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
x, t, X = T._sample(N, L, d, 32)
g = torch.Generator().manual_seed(33)
start = torch.randn(N, dtype=torch.float64, generator=g).requires_grad_(True)
ubar = torch.randn(N, L, dtype=torch.float64, generator=g)
x64 = x.double().requires_grad_(True)
Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
u_ref = R.u_net(th, cfg, Xd, start)
order = [k for k in T.U_ORDER if k in theta]
grads = torch.autograd.grad((u_ref * ubar).sum(), [x64, start] + [th[k] for k in order], allow_unused=True)
blob = torch.cat([(theta[k] if k in theta else torch.zeros(Kk * Kk if k == 'Wh' else Kk, dtype=torch.float64)).reshape(-1) for k in T.U_ORDER]).cuda()
xT, tc, sc = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda()
mid = KN.method_id(solver)
u, Y = KN.ode_fwd(xT, tc, sc, blob, mid, Hh, Kk, m)
print('u err', float((u.t().cpu() - u_ref.detach()).abs().max()))
ub = ubar.t().contiguous().cuda()
def rel(a, b): return float((a.cpu() - b).abs().max()) / float(b.abs().max())
for wx, wp in ((True, False), (True, True), (False, True)):
    gx, gs, slab = KN.ode_bwd(xT, tc, sc, blob, Y, ub, mid, Hh, Kk, m, want_x=wx, want_params=wp)
    msg = 'want_x %s want_params %s:' % (wx, wp)
    if wx:
        e = (gx.t().cpu() - grads[0]).abs()
        msg += ' gx %.2e (worst path %d dim %d) gs %.2e' % (rel(gx.t(), grads[0]), int(e.max(1).values.argmax()), int(e.max(0).values.argmax()), rel(gs, grads[1]))
        bad = (e.max(1).values > 1e-9).nonzero().view(-1).tolist()
        msg += ' bad paths %s' % bad
    if wp:
        flat, off = KN.slab_sum(slab).cpu(), 0
        for k in T.U_ORDER:
            n = theta[k].numel() if k in theta else (Kk * Kk if k == 'Wh' else Kk)
            if k in theta:
                msg += ' %s %.1e' % (k, rel(flat[off:off + n].view(theta[k].shape), grads[2 + order.index(k)]))
            off += n
    print(msg)
