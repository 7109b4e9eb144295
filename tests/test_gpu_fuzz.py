"""Seeded random shapes: the stepper (forward, both sweeps, with and without the activation store) and the test network
(forward with tangent, fused gradient, record, reverse from the record) against the oracle at sizes the fixed cases of
test_gpu_kernels.py do not hit -- ragged last tiles, one path, two time points, every depth, every fixed-grid method,
grid caps that force the ticket queues.  Same tolerances as there (float64 on both sides)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_kernels import F64, U_ORDER, V_ORDER, _blob, _close, _sample, _setup   # noqa: E402


def _cases(n, seed):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        out.append(dict(N=rnd.choice([1, 2, 15, 16, 17, 31, 33, 47, 64, 65, 100, 129, 200]), L=rnd.randint(2, 9), d=rnd.randint(1, 30),
                        m=rnd.randint(1, 8), q=rnd.randint(1, 12), solver=rnd.choice(['euler', 'midpoint', 'rk4']),
                        width=rnd.choice([(20, 10), (32, 12)]), W=rnd.choice([50, 64]), seed=1000 + i))
    return out


@pytest.mark.parametrize('c', _cases(16, 7), ids=lambda c: 'N%d-L%d-d%d-m%d-%s-H%d' % (c['N'], c['L'], c['d'], c['m'], c['solver'], c['width'][0]))
def test_stepper_random_shapes(c):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d, m, solver, (H, K) = c['N'], c['L'], c['d'], c['m'], c['solver'], c['width']
    cfg = {'alpha': 1.0, 'u_layers': m, 'u_hidden_dim': H, 'u_hidden_hidden_dim': K, 'v_layers': 2, 'v_hidden_dim': 50,
           'n1': 1, 'n2': 1, 'u_rate': 0.01, 'v_rate': 0.01, 'min_steps': 5, 'adjoint': False, 'solver': solver}
    torch.manual_seed(c['seed'])
    theta, _ = R.init_parameters(cfg, _setup(d, 2))
    for p in theta.values():
        if p.dim() == 1:
            p.copy_(0.3 * torch.randn_like(p))
    order = [k for k in U_ORDER if k in theta]
    th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
    x, t, _ = _sample(N, L, d, c['seed'] + 1)
    g = torch.Generator().manual_seed(c['seed'] + 2)
    start = torch.randn(N, dtype=F64, generator=g).requires_grad_(True)
    ubar = torch.randn(N, L, dtype=F64, generator=g)
    x64 = x.double().requires_grad_(True)
    Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
    u_ref = R.u_net(th, cfg, Xd, start)
    wrt = [x64, start] + [th[k] for k in order]
    grads = torch.autograd.grad((u_ref * ubar).sum(), wrt, allow_unused=True)     # (m = 1: no tied hidden layer in the graph)
    grads = [torch.zeros_like(w_) if g_ is None else g_ for g_, w_ in zip(grads, wrt)]
    # blob in the kernels' order; a field without hidden layer (m = 1) has no Wh / Wh_b: zeros in their place
    pieces = [theta[k].reshape(-1) if k in theta else torch.zeros(K * K if k == 'Wh' else K, dtype=F64) for k in U_ORDER]
    blob = torch.cat(pieces).cuda()
    assert blob.numel() == KN.theta_size(d, H, K)
    xT, tc, sc, mid = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda(), KN.method_id(solver)
    rows = KN.ode_act_rows(mid, H, K, m)
    ref = torch.cat([(grads[2 + order.index(k)].reshape(-1) if k in order else torch.zeros(K * K if k == 'Wh' else K, dtype=F64))
                     for k in U_ORDER])
    for with_act in ([False, True] if rows else [False]):
        u, Y = torch.empty(L, N, dtype=F64).cuda(), torch.empty(L, H, N, dtype=F64).cuda()
        job = dict(xT=xT, start=sc, u=u, Y=Y)
        if with_act:
            job['act'] = torch.empty(L - 1, rows, KN.ode_act_cols(N), dtype=F64).cuda()
        KN.ode_fwd_multi([job], tc, blob, mid, H, K, m)
        _close(u.t(), u_ref, 1e-12, 'u')
        ub = ubar.t().contiguous().cuda()
        gx, gs = torch.empty(d, N, dtype=F64).cuda(), torch.empty(N, dtype=F64).cuda()
        KN.ode_bwd_multi([dict(job, ubar=ub, gx=gx, gs=gs)], tc, blob, mid, H, K, m, want_x=True, want_params=False)
        _close(gx.t(), grads[0], 1e-10, 'gx (x-only, store %s)' % with_act); _close(gs, grads[1], 1e-10, 'gs')
        slab = torch.empty(KN.ode_bwd_slabs(N), blob.numel(), dtype=F64).cuda()
        gx2, gs2 = torch.empty_like(gx), torch.empty_like(gs)
        KN.ode_bwd_multi([dict(job, ubar=ub, gx=gx2, gs=gs2, gslab=slab)], tc, blob, mid, H, K, m, want_x=True, want_params=True)
        _close(gx2.t(), grads[0], 1e-10, 'gx (param sweep, store %s)' % with_act); _close(gs2, grads[1], 1e-10, 'gs')
        _close(KN.slab_sum(slab).cpu(), ref, 1e-10, 'theta gradient (store %s)' % with_act)


@pytest.mark.parametrize('c', _cases(12, 11), ids=lambda c: 'N%d-L%d-d%d-q%d-W%d' % (c['N'], c['L'], c['d'], c['q'], c['W']))
def test_test_network_random_shapes(c):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d, q, Ww = c['N'], c['L'], c['d'], c['q'], c['W']
    cfg = {'alpha': 1.0, 'u_layers': 2, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': q, 'v_hidden_dim': Ww,
           'n1': 1, 'n2': 1, 'u_rate': 0.01, 'v_rate': 0.01, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint'}
    torch.manual_seed(c['seed'])
    _, phi = R.init_parameters(cfg, _setup(d, 2))
    for p in phi.values():
        if p.dim() == 1:
            p.copy_(0.3 * torch.randn_like(p))
    ph = {k: v.clone().requires_grad_(True) for k, v in phi.items()}
    x, t, X = _sample(N, L, d, c['seed'] + 1)
    Xd = X.double().requires_grad_(True)
    vbar = torch.randn(N, L, dtype=F64, generator=torch.Generator().manual_seed(c['seed'] + 2))
    v_ref = R.v_net(ph, cfg, Xd)
    gX = torch.autograd.grad(v_ref.sum(), Xd, retain_graph=True)[0]
    grads = torch.autograd.grad((v_ref * vbar).sum(), [ph[k] for k in V_ORDER])
    xT, tc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER)
    gxv, gtv = torch.empty(d, N, dtype=F64).cuda(), torch.empty(N, dtype=F64).cuda()
    rec = torch.empty(KN.disc_act_rows(Ww, q), KN.disc_act_cols(L * N), dtype=F64).cuda()
    cap = 1 + c['seed'] % 3                                   # 4 .. 12 waves: several rounds of tiles -> ticket queues
    v, vt = KN.disc_fwd(xT, tc, blob, Ww, q, gxv=gxv, gtv=gtv, ngrad=N, act=rec, max_blocks=cap)
    _close(v.t(), v_ref, 1e-12, 'v'); _close(vt.t(), gX[:, :, 0], 1e-11, 'dv/dt')
    _close(gxv.t(), gX[:, 0, 1:], 1e-11, 'nabla_x v(t_0)'); _close(gtv, gX[:, 0, 0], 1e-11, 'dv/dt(t_0)')
    got = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar.t().contiguous().cuda(), Ww, q, act=rec)).cpu()
    _close(got, torch.cat([g_.reshape(-1) for g_ in grads]), 1e-10, 'phi gradient')
