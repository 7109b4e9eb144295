"""Parity of every HIP kernel (called through the C ABI) against the oracle on the same seeded inputs.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
Tolerances: everything is float64 on both sides; the only differences are summation order and tanh()'s last bits,
so values agree to ~1e-12 relative to their scale and gradients to ~1e-10.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, K, W, Q = 20, 10, 50, 9
F64 = torch.float64


def _cfg(m=8, solver='midpoint'):
    return {'alpha': 1e8, 'u_layers': m, 'u_hidden_dim': H, 'u_hidden_hidden_dim': K, 'v_layers': Q, 'v_hidden_dim': W,
            'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': solver}


def _setup(d, L):
    return {'dim': d, 'N_t': L, 'N_r': 1, 'N_b': 1, 'T0': 0, 'T': 1, 'shape_param': [-1, 1]}


def _params(d, m, seed):
    from oracle import refspec as R
    torch.manual_seed(seed)
    theta, phi = R.init_parameters(_cfg(m), _setup(d, 2))
    # non-zero biases so that every bias path is exercised
    for p in list(theta.values()) + list(phi.values()):
        if p.dim() == 1:
            p.copy_(0.3 * torch.randn_like(p))
    return theta, phi


def _blob(p, order):
    return torch.cat([p[k].reshape(-1) for k in order]).cuda()


U_ORDER = ['IL0_w', 'IL0_b', 'IL2_w', 'IL2_b', 'IL4_w', 'IL4_b', 'Win', 'Win_b', 'Wh', 'Wh_b', 'Wo', 'Wo_b', 'FL_w', 'FL_b']
V_ORDER = ['Vin', 'Vin_b', 'Vh', 'Vh_b', 'Vo', 'Vo_b']


def _sample(N, L, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(N, d, generator=g) * 2 - 1).float()
    t, _ = torch.sort(torch.rand(L, generator=g).float())
    t[0], t[-1] = 0.0, 1.0
    X = torch.cat((t.view(1, L, 1).expand(N, L, 1), x.view(N, 1, d).expand(N, L, d)), 2).contiguous()
    return x, t, X


def _close(a, b, tol, what):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-300)
    err = float((a - b).abs().max()) / scale
    assert err < tol, '%s: max rel-to-scale error %.3e (scale %.3e)' % (what, err, scale)


CASES = [(37, 7, 5), (64, 6, 20), (16, 2, 3)]


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('N,L,d', CASES)
def test_ode_forward(N, L, d, solver):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    theta, _ = _params(d, 8, 1)
    x, t, X = _sample(N, L, d, 2)
    start = torch.randn(N, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    u_ref = R.u_net(theta, _cfg(8, solver), X, start)
    u, Y = KN.ode_fwd(x.double().t().contiguous().cuda(), t.double().cuda(), start.double().cuda(), _blob(theta, U_ORDER), KN.method_id(solver), H, K, 8)
    _close(u.t(), u_ref, 1e-12, 'u')
    assert Y.shape == (L, H, N)


@pytest.mark.parametrize('m', [10, 9, 8, 7, 6, 5, 4, 3, 2, 1])
def test_ode_forward_depths(m):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d = 33, 5, 4
    theta, _ = _params(d, m, 5)
    x, t, X = _sample(N, L, d, 6)
    start = torch.randn(N, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    u_ref = R.u_net(theta, _cfg(m), X, start)
    u, _ = KN.ode_fwd(x.double().t().contiguous().cuda(), t.double().cuda(), start.double().cuda(), _blob(theta, U_ORDER), 1, H, K, m)
    _close(u.t(), u_ref, 1e-12, 'u')


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('N,L,d', CASES)
@pytest.mark.parametrize('ones', [False, True])
def test_ode_backward(N, L, d, solver, ones):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    theta, _ = _params(d, 8, 11)
    th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
    x, t, X = _sample(N, L, d, 12)
    g = torch.Generator().manual_seed(13)
    start = torch.randn(N, dtype=torch.float64, generator=g).requires_grad_(True)
    ubar = torch.ones(N, L, dtype=torch.float64) if ones else torch.randn(N, L, dtype=torch.float64, generator=g)
    # x enters only through time slice 0 (src/model.py:99): differentiate w.r.t. a float64 copy of that slice
    x64 = x.double().requires_grad_(True)
    Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
    u_ref = R.u_net(th, _cfg(8, solver), Xd, start)
    grads = torch.autograd.grad((u_ref * ubar).sum(), [x64, start] + [th[k] for k in U_ORDER])
    xT, tc, sc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda(), _blob(theta, U_ORDER)
    mid = KN.method_id(solver)
    u, Y = KN.ode_fwd(xT, tc, sc, blob, mid, H, K, 8)
    ub = None if ones else ubar.t().contiguous().cuda()
    gx, gs, _ = KN.ode_bwd(xT, tc, sc, blob, Y, ub, mid, H, K, 8, want_x=True, want_params=False)
    _close(gx.t(), grads[0], 1e-10, 'gx (x-only sweep)')
    _close(gs, grads[1], 1e-10, 'gs (x-only sweep)')
    gx2, gs2, slab = KN.ode_bwd(xT, tc, sc, blob, Y, ub, mid, H, K, 8, want_x=True, want_params=True)
    _close(gx2.t(), grads[0], 1e-10, 'gx (param sweep)')
    _close(gs2, grads[1], 1e-10, 'gs (param sweep)')
    gth = KN.slab_sum(slab).cpu()
    ref = torch.cat([g_.reshape(-1) for g_ in grads[2:]])
    off = 0
    for k, g_ in zip(U_ORDER, grads[2:]):
        n = g_.numel()
        _close(gth[off:off + n], g_.reshape(-1), 1e-10 * max(1.0, float(ref.abs().max()) / max(float(g_.abs().max()), 1e-300)), 'grad ' + k)
        off += n
    _close(gth, ref, 1e-10, 'theta grad (whole blob)')


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('N,L,d', [(37, 7, 5), (16, 2, 3)])
def test_ode_backward_continuous_adjoint(N, L, d, solver):
    """config['adjoint'] = True (src/model.py:103): xw_ode_bwd mode bit 3 against the oracle's restatement of
    torchdiffeq.odeint_adjoint -- restart from the forward solution at every grid point, one step of the same method on
    the augmented system, no gradient to x through the field.  (Parity unpinned: torchdiffeq is not in the reference tree.)"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    theta, _ = _params(d, 8, 81)
    th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
    x, t, X = _sample(N, L, d, 82)
    g = torch.Generator().manual_seed(83)
    start = torch.randn(N, dtype=F64, generator=g).requires_grad_(True)
    ubar = torch.randn(N, L, dtype=F64, generator=g)
    x64 = x.double().requires_grad_(True)
    Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
    cfg = dict(_cfg(8, solver), adjoint=True)
    u_ref = R.u_net(th, cfg, Xd, start)
    grads = torch.autograd.grad((u_ref * ubar).sum(), [x64, start] + [th[k] for k in U_ORDER], allow_unused=True)
    assert grads[0] is None or float(grads[0].abs().max()) == 0.0        # x is not an input of odeint_adjoint
    # the discrete gradient is a different number (O(dt^p) away): the two modes must not be confused
    g_disc = torch.autograd.grad((R.u_net(th, _cfg(8, solver), Xd, start) * ubar).sum(), [th['Wh']])[0]
    assert float((g_disc - grads[2 + U_ORDER.index('Wh')]).abs().max()) > 1e-6 * float(g_disc.abs().max())
    xT, tc, sc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda(), _blob(theta, U_ORDER)
    mid = KN.method_id(solver)
    u, Y = KN.ode_fwd(xT, tc, sc, blob, mid, H, K, 8)
    _close(u.t(), u_ref, 1e-12, 'u')
    ub = ubar.t().contiguous().cuda()
    gx, gs, _ = KN.ode_bwd(xT, tc, sc, blob, Y, ub, mid, H, K, 8, want_x=True, want_params=False, adjoint=True)
    assert float(gx.abs().max()) == 0.0
    _close(gs, grads[1], 1e-10, 'gs (x-only sweep)')
    gx2, gs2, slab = KN.ode_bwd(xT, tc, sc, blob, Y, ub, mid, H, K, 8, want_x=True, want_params=True, adjoint=True)
    assert float(gx2.abs().max()) == 0.0
    _close(gs2, grads[1], 1e-10, 'gs (param sweep)')
    gth = KN.slab_sum(slab).cpu()
    ref = torch.cat([g_.reshape(-1) for g_ in grads[2:]])
    off = 0
    for k, g_ in zip(U_ORDER, grads[2:]):
        n = g_.numel()
        _close(gth[off:off + n], g_.reshape(-1), 1e-10 * max(1.0, float(ref.abs().max()) / max(float(g_.abs().max()), 1e-300)), 'grad ' + k)
        off += n


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
def test_ode_backward_pollution_and_x_sweep_in_one(solver):
    """mode 7: cotangent = ones + (initial-value term at time index 0).  One sweep must return the parameter gradient of
    that cotangent AND the x / start gradients of the all-ones cotangent (the helper backward of src/loss.py:55),
    for two groups in one launch, the second without x outputs."""
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d = 37, 6, 20
    theta, _ = _params(d, 8, 31)
    x, t, _ = _sample(N, L, d, 32)
    g = torch.Generator().manual_seed(33)
    start = torch.randn(N, dtype=F64, generator=g)
    ubar = torch.ones(L, N, dtype=F64)
    ubar[0] += 50.0 * torch.randn(N, dtype=F64, generator=g)
    xT, tc, sc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), start.cuda(), _blob(theta, U_ORDER)
    mid = KN.method_id(solver)
    u, Y = KN.ode_fwd(xT, tc, sc, blob, mid, H, K, 8)
    ub = ubar.cuda()
    gx1, gs1, _ = KN.ode_bwd(xT, tc, sc, blob, Y, None, mid, H, K, 8, want_x=True, want_params=False)
    _, _, slab1 = KN.ode_bwd(xT, tc, sc, blob, Y, ub, mid, H, K, 8, want_x=False, want_params=True)
    gx, gs = torch.empty_like(gx1), torch.empty_like(gs1)
    slab, slab_b = torch.empty_like(slab1), torch.empty_like(slab1)
    u2 = torch.empty_like(u)
    jobs = [dict(xT=xT, start=sc, u=u2, Y=Y, ubar=ub, gx=gx, gs=gs, gslab=slab),
            dict(xT=xT, start=sc, u=u2, Y=Y, ubar=ub, gslab=slab_b)]
    KN.ode_bwd_multi(jobs, tc, blob, mid, H, K, 8, want_x=True, want_params=True, x_cot_ones=True)
    _close(gx, gx1, 1e-12, 'gx of the ones cotangent')
    _close(gs, gs1, 1e-11, 'gs of the ones cotangent')
    assert torch.equal(slab, slab1) and torch.equal(slab_b, slab1)


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
def test_ode_backward_from_stored_activations(solver):
    """the sweeps read the stage activations the forward stored (XwOdeFwdJob.act) instead of re-evaluating the field:
    same gradients as the recomputing sweeps (rk4 ignores the store)"""
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d = 37, 6, 20
    theta, _ = _params(d, 8, 41)
    x, t, _ = _sample(N, L, d, 42)
    g = torch.Generator().manual_seed(43)
    start = torch.randn(N, dtype=F64, generator=g)
    ub = torch.randn(L, N, dtype=F64, generator=g).cuda()
    xT, tc, sc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), start.cuda(), _blob(theta, U_ORDER)
    mid = KN.method_id(solver)
    rows = KN.ode_act_rows(mid, H, K, 8)
    if (H, K) == (20, 10):
        assert rows == {'euler': 80 + 2, 'midpoint': 180 + 4, 'rk4': 0}[solver]      # layer inputs + 2 rows of mask words per stage
    else:
        assert (rows > 0) == (solver != 'rk4')
    u0, Y0 = KN.ode_fwd(xT, tc, sc, blob, mid, H, K, 8)
    gx0, gs0, slab0 = KN.ode_bwd(xT, tc, sc, blob, Y0, ub, mid, H, K, 8, want_x=True, want_params=True)
    u, Y = torch.empty_like(u0), torch.empty_like(Y0)
    act = torch.full((L - 1, max(rows, 1), KN.ode_act_cols(N)), float('nan'), dtype=F64, device='cuda')
    job = dict(xT=xT, start=sc, u=u, Y=Y, act=act if rows else None)
    KN.ode_fwd_multi([job], tc, blob, mid, H, K, 8)
    assert torch.equal(u, u0) and torch.equal(Y, Y0)
    if rows:
        assert torch.isfinite(act).all()                     # every row of every step was written
    gx, gs, slab = torch.empty_like(gx0), torch.empty_like(gs0), torch.empty_like(slab0)
    KN.ode_bwd_multi([dict(job, ubar=ub, gx=gx, gs=gs, gslab=slab)], tc, blob, mid, H, K, 8, want_x=True, want_params=True)
    _close(gx, gx0, 1e-12, 'gx'); _close(gs, gs0, 1e-12, 'gs')
    _close(KN.slab_sum(slab), KN.slab_sum(slab0), 1e-12, 'theta gradient')
    if rows:
        # a store that will only serve a sweep without weight gradients: tanh rows + ReLU mask words (XwOdeFwdJob.act_x_only)
        actx = torch.full_like(act, float('nan'))
        KN.ode_fwd_multi([dict(job, act=actx)], tc, blob, mid, H, K, 8, act_x_only=True)
        assert torch.equal(u, u0) and int(torch.isfinite(actx).sum()) < int(torch.isfinite(act).sum()) // 3
        gx1, gs1 = torch.empty_like(gx0), torch.empty_like(gs0)
        KN.ode_bwd_multi([dict(job, act=actx, ubar=ub, gx=gx1, gs=gs1)], tc, blob, mid, H, K, 8, want_x=True, want_params=False)
        gx2, gs2 = torch.empty_like(gx0), torch.empty_like(gs0)
        KN.ode_bwd_multi([dict(job, ubar=ub, gx=gx2, gs=gs2)], tc, blob, mid, H, K, 8, want_x=True, want_params=False)
        assert torch.equal(gx1, gx2) and torch.equal(gs1, gs2)
        _close(gx1, gx0, 1e-12, 'gx (x-only store)'); _close(gs1, gs0, 1e-12, 'gs (x-only store)')


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
def test_ode_sweep_with_residual_cotangents(solver):
    """XwOdeBwdJob.res_*: the cotangent base + coef (u - ref) formed inside the sweep (initial-value penalty: ref[n], l = 0
    only; boundary penalty: ref[l][n] at every l) must give the outputs of the same cotangent handed over as a stored array
    (to rounding: the kernel forms base + coef * r with one fused multiply-add) -- with and without the activation store,
    with weight gradients (duo sweep) and with x outputs."""
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d = 300, 6, 5
    theta, _ = _params(d, 8, 71)
    x, t, _ = _sample(N, L, d, 72)
    g = torch.Generator().manual_seed(73)
    start = torch.randn(N, dtype=torch.float64, generator=g).cuda()
    href = torch.randn(N, dtype=torch.float64, generator=g).cuda()
    gref = torch.randn(L, N, dtype=torch.float64, generator=g).cuda()
    xT, tc, blob, mid = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(theta, U_ORDER), KN.method_id(solver)
    rows = KN.ode_act_rows(mid, H, K, 8)
    for with_act in ([False, True] if rows else [False]):
        u, Y = torch.empty(L, N, dtype=F64).cuda(), torch.empty(L, H, N, dtype=F64).cuda()
        job = dict(xT=xT, start=start, u=u, Y=Y)
        if with_act:
            job['act'] = torch.empty(L - 1, rows, KN.ode_act_cols(N), dtype=F64).cuda()
        KN.ode_fwd_multi([job], tc, blob, mid, H, K, 8)
        for first_only, ref, coef, base in ((True, href, 0.37, 1.0), (False, gref, -1.3, 0.0)):
            ubar = torch.full((L, N), base, dtype=F64).cuda()
            if first_only:
                ubar[0] += coef * (u[0] - ref)
            else:
                ubar += coef * (u - ref)
            res = dict(u=u, ref=ref, coef=coef, base=base, first_only=first_only)
            outs = []
            for kw in (dict(ubar=ubar), dict(res=res)):
                gx, gs = torch.empty(d, N, dtype=F64).cuda(), torch.empty(N, dtype=F64).cuda()
                slab = torch.empty(KN.ode_bwd_slabs(N), blob.numel(), dtype=F64).cuda()
                KN.ode_bwd_multi([dict(job, gx=gx, gs=gs, gslab=slab, **kw)], tc, blob, mid, H, K, 8, want_x=True, want_params=True,
                                 x_cot_ones=first_only)
                slab2 = torch.empty_like(slab)
                KN.ode_bwd_multi([dict(job, gslab=slab2, **kw)], tc, blob, mid, H, K, 8, want_x=False, want_params=True)
                outs.append((gx, gs, slab, slab2))
            for a_, b_, what in zip(outs[0], outs[1], ('gx', 'gs', 'slab (with x outputs)', 'slab')):
                _close(a_, b_, 1e-13, '%s: %s, store %s, first_only %s' % (what, solver, with_act, first_only))
        # the weak form's dI/du (res_first_only = 2) against xw_gen_cotangents' basis B handed over as a stored array:
        # c = kappa u, and tabulated c, dc/du with a per-point weight
        v_ = torch.randn(L, N, dtype=F64, generator=g).cuda()
        w_n, w_ln = torch.rand(N, dtype=F64, generator=g).cuda(), torch.rand(L, N, dtype=F64, generator=g).cuda()
        c_, cp_ = torch.randn(L, N, dtype=F64, generator=g).cuda(), torch.randn(L, N, dtype=F64, generator=g).cuda()
        Vol, Ng = 3.7, 1234.0
        for w_, ctab in ((w_n, False), (w_ln, True)):
            ubarB = torch.empty(L, N, dtype=F64).cuda()
            KN.gen_cotangents(u, v_, w_, href, Vol, Ng, 1.0, None, ubarB, c=c_ if ctab else None, cp=cp_ if ctab else None,
                              ckappa=0.0 if ctab else -0.8)
            resB = dict(u=u, ref=v_, coef=Vol / Ng / L, base=Vol / Ng,
                        weak=dict(w=w_, c=c_ if ctab else None, cp=cp_ if ctab else None, ckappa=0.0 if ctab else -0.8))
            outs = []
            for kw in (dict(ubar=ubarB), dict(res=resB)):
                slab = torch.empty(KN.ode_bwd_slabs(N), blob.numel(), dtype=F64).cuda()
                KN.ode_bwd_multi([dict(job, gslab=slab, **kw)], tc, blob, mid, H, K, 8, want_x=False, want_params=True)
                outs.append(slab)
            _close(outs[1], outs[0], 1e-13, 'slab, dI/du formed in the sweep (%s, store %s, tabulated c %s)' % (solver, with_act, ctab))
    with pytest.raises(Exception):
        KN.ode_bwd_multi([dict(job, gslab=slab, ubar=ubar, res=res)], tc, blob, mid, H, K, 8, want_x=False, want_params=True)


@pytest.mark.parametrize('solver', ['euler', 'midpoint'])
@pytest.mark.parametrize('Hh,Kk,m,N,L,d', [(20, 10, 8, 37, 7, 5), (20, 10, 8, 64, 6, 20), (20, 10, 8, 16, 2, 3), (20, 10, 8, 1, 3, 1),
                                           (20, 10, 3, 50, 5, 6), (20, 10, 1, 21, 4, 6), (20, 10, 8, 300, 4, 70),
                                           (32, 12, 8, 37, 6, 5), (32, 12, 2, 19, 3, 21),
                                           (20, 10, 10, 37, 5, 5), (20, 10, 9, 20, 3, 4), (32, 12, 10, 19, 3, 6)])   # (deepest compiled fields)
def test_ode_narrow_tile_sweeps(Hh, Kk, m, N, L, d, solver):
    """xw_ode_bwd mode bit 4 (csrc/xw_ode_n4.h): the sweeps with a 16-path tile spread over four waves of 4 paths x 16 rows
    -- x, start and every weight gradient against the oracle's autograd (1e-10), and against the 16-path sweeps of the same
    store (1e-12: only the summation order of the weight gradients differs); x-only sweep from the reduced store; the fused
    pollution + nabla_x u form; two jobs in one launch; residual cotangents"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    cfg = dict(_cfg(m, solver), u_hidden_dim=Hh, u_hidden_hidden_dim=Kk)
    torch.manual_seed(131)
    theta, _ = R.init_parameters(cfg, _setup(d, 2))
    for p in theta.values():
        if p.dim() == 1:
            p.copy_(0.3 * torch.randn(p.shape, dtype=torch.float64))
    th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
    x, t, X = _sample(N, L, d, 132)
    g = torch.Generator().manual_seed(133)
    start = torch.randn(N, dtype=torch.float64, generator=g).requires_grad_(True)
    ubar = torch.randn(N, L, dtype=torch.float64, generator=g)
    x64 = x.double().requires_grad_(True)
    Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
    u_ref = R.u_net(th, cfg, Xd, start)
    order = [k for k in U_ORDER if k in theta]
    grads = torch.autograd.grad((u_ref * ubar).sum(), [x64, start] + [th[k] for k in order], allow_unused=True)
    grads = [g_ if g_ is not None else torch.zeros_like(p_) for g_, p_ in zip(grads, [x64, start] + [th[k] for k in order])]
    blob = torch.cat([(theta[k] if k in theta else torch.zeros(Kk * Kk if k == 'Wh' else Kk, dtype=torch.float64)).reshape(-1)
                      for k in U_ORDER]).cuda()
    xT, tc, sc = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda()
    ub = ubar.t().contiguous().cuda()
    mid = KN.method_id(solver)
    rows = KN.ode_act_rows(mid, Hh, Kk, m)
    new = lambda *shape: torch.full(shape, float('nan'), dtype=F64, device='cuda')      # noqa: E731
    job = dict(xT=xT, start=sc, u=new(L, N), Y=new(L, Hh, N), act=new(L - 1, rows, KN.ode_act_cols(N)))
    KN.ode_fwd_multi([job], tc, blob, mid, Hh, Kk, m)
    nsl, P = KN.ode_bwd_slabs(N), blob.numel()

    def sweep(narrow, want_x=True, want_params=True, j=job, **kw):
        gx, gs, slab = new(d, N), new(N), new(nsl, P)
        KN.ode_bwd_multi([dict(j, ubar=ub, gx=gx, gs=gs, gslab=slab)], tc, blob, mid, Hh, Kk, m, want_x=want_x, want_params=want_params,
                         narrow=narrow, **kw)
        return gx, gs, slab
    gx, gs, slab = sweep(True)
    gxw, gsw, slabw = sweep(False)
    _close(gx.t(), grads[0], 1e-10, 'gx'); _close(gs, grads[1], 1e-10, 'gs')
    _close(gx, gxw, 1e-12, 'gx vs the 16-path sweep'); _close(gs, gsw, 1e-12, 'gs vs the 16-path sweep')
    assert torch.isfinite(slab).all()
    flat, flatw, off = KN.slab_sum(slab).cpu(), KN.slab_sum(slabw).cpu(), 0
    _close(flat, flatw, 1e-12, 'theta gradient vs the 16-path sweep')
    for k in U_ORDER:
        n = theta[k].numel() if k in theta else (Kk * Kk if k == 'Wh' else Kk)
        if k in theta:
            _close(flat[off:off + n].view(theta[k].shape), grads[2 + order.index(k)], 1e-10, 'grad ' + k)
        off += n
    # the forward pass on narrow tiles (XwOdeFwdJob.narrow): same u, checkpoints and activation record (compared through the
    # sweeps that read it: the 16-path sweep from the narrow store, the narrow sweep from the narrow store)
    _close(job['u'].t(), u_ref.detach(), 1e-12, 'u (16-path forward)')
    jn = dict(job, u=new(L, N), Y=new(L, Hh, N), act=new(L - 1, rows, KN.ode_act_cols(N)))
    KN.ode_fwd_multi([jn], tc, blob, mid, Hh, Kk, m, narrow=True)
    _close(jn['u'].t(), u_ref.detach(), 1e-12, 'u (narrow forward)')
    _close(jn['Y'], job['Y'], 1e-12, 'checkpoints (narrow forward)')
    ncol = KN.ode_act_cols(N)
    written = torch.isfinite(job['act'])
    assert torch.equal(torch.isfinite(jn['act']), written), 'the narrow forward fills the same slots of the record'
    floats = jn['act'][:, :rows - 2 * (1 if solver == 'euler' else 2)]            # (the mask words behind them are not doubles)
    _close(torch.nan_to_num(floats), torch.nan_to_num(job['act'][:, :floats.shape[1]]), 1e-12, 'activation record (narrow forward)')
    for narrow in (False, True):
        gxn, gsn, slabn = sweep(narrow, j=jn)
        _close(gxn, gxw, 1e-12, 'gx from the narrow store'); _close(gsn, gsw, 1e-12, 'gs from the narrow store')
        _close(KN.slab_sum(slabn).cpu(), flatw, 1e-12, 'theta gradient from the narrow store')
    jnx = dict(jn, act=new(L - 1, rows, KN.ode_act_cols(N)))
    KN.ode_fwd_multi([jnx], tc, blob, mid, Hh, Kk, m, act_x_only=True, narrow=True)
    gxn, gsn, _ = sweep(True, want_params=False, j=jnx)
    _close(gxn, gxw, 1e-12, 'gx (reduced narrow store)'); _close(gsn, gsw, 1e-12, 'gs (reduced narrow store)')
    # without weight gradients, from a store that only holds the tanh rows and the mask words
    jx = dict(job, act=new(L - 1, rows, KN.ode_act_cols(N)))
    KN.ode_fwd_multi([jx], tc, blob, mid, Hh, Kk, m, act_x_only=True)
    gx1, gs1, _ = sweep(True, want_params=False, j=jx)
    _close(gx1, gxw, 1e-12, 'gx (x-only, reduced store)'); _close(gs1, gsw, 1e-12, 'gs (x-only, reduced store)')
    # weight gradients only
    _, _, slab2 = sweep(True, want_x=False)
    assert torch.equal(slab2, slab)
    # the generator sub-step's fused form: x outputs for the all-ones cotangent, parameters for ubar (== 1 behind l = 0)
    ub1 = torch.ones(L, N, dtype=F64, device='cuda')
    ub1[0] = ub[0]
    a = [new(d, N), new(N), new(nsl, P)]
    b = [new(d, N), new(N), new(nsl, P)]
    for out, narrow in ((a, True), (b, False)):
        KN.ode_bwd_multi([dict(job, ubar=ub1, gx=out[0], gs=out[1], gslab=out[2])], tc, blob, mid, Hh, Kk, m, want_x=True,
                         want_params=True, x_cot_ones=True, narrow=narrow)
    for p_, q_, what in zip(a[:2], b[:2], ('gx (ones)', 'gs (ones)')):
        _close(p_, q_, 1e-12, what)
    _close(KN.slab_sum(a[2]), KN.slab_sum(b[2]), 1e-12, 'theta gradient (fused form)')
    # two jobs in one launch (the second with its own paths), one of them with a residual cotangent formed in the sweep
    N2 = max(N // 2, 1)
    x2, _, _ = _sample(N2, L, d, 134)
    job2 = dict(xT=x2.double().t().contiguous().cuda(), start=torch.randn(N2, dtype=F64, generator=g).cuda(), u=new(L, N2),
                Y=new(L, Hh, N2), act=new(L - 1, rows, KN.ode_act_cols(N2)))
    KN.ode_fwd_multi([job, job2], tc, blob, mid, Hh, Kk, m)
    gref = torch.randn(L, N2, dtype=F64, generator=g).cuda()
    res = dict(u=job2['u'], ref=gref, coef=-1.3, base=0.25, first_only=False)
    outs = []
    for narrow in (True, False):
        s1, s2 = new(nsl, P), new(KN.ode_bwd_slabs(N2), P)
        KN.ode_bwd_multi([dict(job, ubar=ub, gslab=s1), dict(job2, res=res, gslab=s2)], tc, blob, mid, Hh, Kk, m, want_x=False,
                         want_params=True, narrow=narrow)
        outs.append((KN.slab_sum(s1), KN.slab_sum(s2)))
    assert torch.equal(outs[0][0].cpu(), flat)
    _close(outs[0][1], outs[1][1], 1e-12, 'second job of the launch, residual cotangent')
    with pytest.raises(Exception):                                  # no store, no narrow sweep
        KN.ode_bwd_multi([dict(job, act=None, ubar=ub, gx=new(d, N), gs=new(N))], tc, blob, mid, Hh, Kk, m, want_x=True,
                         want_params=False, narrow=True)


def test_slab_sums_against_torch():
    """xw_slab_sum / xw_slab_sum2 (fixed-tree reductions of the per-wave gradient slabs) against torch.sum, at slab counts
    below, at and above the 64 groups of a block, with accumulation"""
    from xnode_wan_pde_solver_amd import kernels as KN
    g = torch.Generator().manual_seed(91)
    for ns, P in ((1, 7), (63, 1551), (64, 16), (65, 17), (512, 1551), (256, 3701)):
        S = torch.randn(ns, P, dtype=F64, generator=g).cuda()
        out = KN.slab_sum(S)
        _close(out, S.sum(0), 1e-13, 'slab_sum %d x %d' % (ns, P))
        out2 = KN.slab_sum(S, out=out.clone(), accumulate=True)
        _close(out2, 2 * S.sum(0), 1e-13, 'slab_sum accumulate')
        T = torch.randn(max(ns // 2, 1), P, dtype=F64, generator=g).cuda()
        a, b = torch.empty(P, dtype=F64).cuda(), torch.empty(P, dtype=F64).cuda()
        KN.slab_sum2(S, a, T, b)
        assert torch.equal(a, out) and torch.equal(b, KN.slab_sum(T))


@pytest.mark.parametrize('N,L,d', CASES + [(50, 3, 70)])
def test_disc_forward_and_time_tangent(N, L, d):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    _, phi = _params(d, 8, 21)
    x, t, X = _sample(N, L, d, 22)
    Xd = X.double().requires_grad_(True)
    v_ref = R.v_net(phi, _cfg(), Xd)
    gX = torch.autograd.grad(v_ref.sum(), Xd)[0]
    v, vt = KN.disc_fwd(x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER), W, Q)
    _close(v.t(), v_ref, 1e-12, 'v')
    _close(vt.t(), gX[:, :, 0], 1e-11, 'dv/dt')
    # point mode: arbitrary per-point times
    tpp = torch.rand(N, generator=torch.Generator().manual_seed(23)).float()
    Xp = torch.cat((tpp.view(N, 1), x), 1).double()
    v1, _ = KN.disc_fwd(x.double().t().contiguous().cuda(), None, _blob(phi, V_ORDER), W, Q, tpp=tpp.double().cuda())
    _close(v1[0], R.v_net(phi, _cfg(), Xp), 1e-12, 'v (point mode)')
    # fused input gradient at the leading points (first time index), with a reduced grid
    gxv = torch.empty(d, N, dtype=torch.float64).cuda()
    gtv = torch.empty(N, dtype=torch.float64).cuda()
    v2, vt2 = KN.disc_fwd(x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER), W, Q, gxv=gxv, gtv=gtv,
                          ngrad=N, max_blocks=3)
    _close(v2.t(), v_ref, 1e-12, 'v (with fused gradient)')
    _close(gxv.t(), gX[:, 0, 1:], 1e-11, 'fused nabla_x v at t0')
    _close(gtv, gX[:, 0, 0], 1e-11, 'fused dv/dt at t0')


def test_disc_time_tangent_at_an_exact_zero_preactivation():
    """relu'(0) = 0 as in torch: at (t, x) = (0, 0) with zero biases every first-layer pre-activation is exactly +0 while its
    t-tangent Vin[:, 0] is not -- dv/dt there must be the gated one (0 through the dead units), not the right derivative"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    d, N, L = 4, 16, 3
    torch.manual_seed(5)
    _, phi = R.init_parameters(_cfg(), _setup(d, 2))             # (biases are zero at initialisation, src/model.py:12-15)
    x = torch.zeros(N, d)
    x[1:] = torch.rand(N - 1, d) * 2 - 1
    t = torch.tensor([0.0, 0.4, 1.0])
    X = torch.cat((t.view(1, L, 1).expand(N, L, 1), x.view(N, 1, d).expand(N, L, d)), 2).contiguous().double().requires_grad_(True)
    v_ref = R.v_net(phi, _cfg(), X)
    dv = torch.autograd.grad(v_ref.sum(), X)[0][:, :, 0]
    v, vt = KN.disc_fwd(x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER), W, Q)
    _close(v.t(), v_ref.detach(), 1e-12, 'v')
    assert float(dv[0, 0]) == 0.0 and float(vt[0, 0]) == 0.0, (float(dv[0, 0]), float(vt[0, 0]))
    _close(vt.t(), dv, 1e-11, 'dv/dt')


@pytest.mark.parametrize('Ww', [64, 96, 100, 128, 51 + 64])      # (64, 96, 128: MFMA containers; 100, 115 handed to the C ABI as they are: the generic path)
@pytest.mark.parametrize('N,L,d,q', [(37, 7, 5, 9), (64, 6, 20, 4), (100, 3, 70, 1), (16, 2, 3, 12), (700, 9, 6, 9),
                                     (1100, 32, 6, 2)])     # (the last one: more 64-point groups than blocks -- grid-stride)
def test_disc_kernels_at_width_64(N, L, d, q, Ww):
    """the second compiled test-network width (64 = four full MFMA row tiles, container of v_hidden_dim 51..64): forward,
    d/dt tangent, fused input gradient, record and the reverse from the record (dVh.b summed on the vector ALU: no padding
    row left for the ones-row trick) against the oracle at v_hidden_dim = 64.  No recomputing reverse kernels at this width."""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    if Ww not in (64, 96, 128) and N * L > 1000:
        pytest.skip('the generic path is slow: small cases only')
    cfg = dict(_cfg(), v_hidden_dim=Ww, v_layers=q)
    torch.manual_seed(61)
    _, phi = R.init_parameters(cfg, _setup(d, 2))
    for p_ in phi.values():
        if p_.dim() == 1:
            p_.copy_(0.3 * torch.randn_like(p_))
    ph = {k: v_.clone().requires_grad_(True) for k, v_ in phi.items()}
    x, t, X = _sample(N, L, d, 62)
    Xd = X.double().requires_grad_(True)
    vbar = torch.randn(N, L, dtype=torch.float64, generator=torch.Generator().manual_seed(63))
    v_ref = R.v_net(ph, cfg, Xd)
    gX = torch.autograd.grad(v_ref.sum(), Xd, retain_graph=True)[0]
    grads = torch.autograd.grad((v_ref * vbar).sum(), [ph[k] for k in V_ORDER])
    xT, tc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER)
    # (round 6: a 128-wide MFMA container serves v_hidden_dim 65..128; the kernels are called at Ww itself here)
    assert blob.numel() == KN.phi_size(d, Ww) and KN.disc_container(51) == 64 and KN.disc_container(64) == 64
    assert KN.disc_container(65) == 96 and KN.disc_container(97) == 128 and KN.disc_container(Ww) == (Ww if Ww in (64, 96) else 128)
    assert not KN.disc_generic(96) and not KN.disc_generic(128)
    gxv, gtv = torch.empty(d, N, dtype=torch.float64).cuda(), torch.empty(N, dtype=torch.float64).cuda()
    rec = torch.empty(KN.disc_act_rows(Ww, q), KN.disc_act_cols(L * N), dtype=torch.float64).cuda()
    v, vt = KN.disc_fwd(xT, tc, blob, Ww, q, gxv=gxv, gtv=gtv, ngrad=N, act=rec, max_blocks=5)
    _close(v.t(), v_ref, 1e-12, 'v'); _close(vt.t(), gX[:, :, 0], 1e-11, 'dv/dt')
    _close(gxv.t(), gX[:, 0, 1:], 1e-11, 'fused nabla_x v at t0'); _close(gtv, gX[:, 0, 0], 1e-11, 'fused dv/dt at t0')
    got = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar.t().contiguous().cuda(), Ww, q, act=rec)).cpu()
    ref = torch.cat([g_.reshape(-1) for g_ in grads])
    off = 0
    for k, g_ in zip(V_ORDER, grads):
        n_ = g_.numel()
        _close(got[off:off + n_], ref[off:off + n_], 1e-10, 'grad ' + k)
        off += n_
    # without a record the Python layer stores one first (kernels.disc_bwd); the raw entry point refuses
    got2 = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar.t().contiguous().cuda(), Ww, q)).cpu()
    _close(got2, got, 1e-13, 'gradient via an implicit record')
    gx2, gt2 = KN.disc_gradx(xT, tc, blob, Ww, q)
    _close(gx2.t(), gX[:, 0, 1:], 1e-11, 'disc_gradx'); _close(gt2, gX[:, 0, 0], 1e-11, 'disc_gradx dt')
    from xnode_wan_pde_solver_amd._lib import lib
    assert lib.xw_disc_gradx(xT.data_ptr(), tc.data_ptr(), None, blob.data_ptr(), None, N, d, Ww, 9, gxv.data_ptr(), gtv.data_ptr(), None) == -1   # XW_E_DIMS
    # point mode (every point its own time) gives the same values at the same points
    tpp = tc.view(L, 1).expand(L, N).reshape(-1).contiguous()
    xp = xT.unsqueeze(1).expand(d, L, N).reshape(d, L * N).contiguous()
    vp, vtp = KN.disc_fwd(xp, None, blob, Ww, q, tpp=tpp)
    _close(vp.view(L, N), v, 1e-13, 'point mode v'); _close(vtp.view(L, N), vt, 1e-13, 'point mode dv/dt')


@pytest.mark.parametrize('Ww,q', [(50, 9), (64, 4), (96, 2), (128, 3)])
@pytest.mark.parametrize('N,L,d', [(37, 7, 5), (64, 6, 20), (100, 3, 70), (300, 5, 100), (1100, 32, 50)])
def test_disc_forward_with_the_hoisted_x_projection(N, L, d, Ww, q):
    """xw_disc_xproj + xw_disc_fwd_xproj (the input layer's spatial columns applied once per path instead of once per point)
    against the oracle and against the plain launch: values, d/dt, fused input gradient, and the gradient from its record"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    from xnode_wan_pde_solver_amd._lib import XnwanError
    cfg = dict(_cfg(), v_hidden_dim=Ww, v_layers=q)
    torch.manual_seed(71)
    _, phi = R.init_parameters(cfg, _setup(d, 2))
    for p_ in phi.values():
        if p_.dim() == 1:
            p_.copy_(0.3 * torch.randn_like(p_))
    x, t, X = _sample(N, L, d, 72)
    xT, tc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER)
    xp = KN.disc_xproj(xT, blob, Ww)
    Vin, b = phi['Vin'].double(), phi['Vin_b'].double()
    ref_p = x.double() @ Vin[:, 1:].t() + b
    _close(xp[:Ww].t(), ref_p, 1e-13, 'x projection')
    assert float(xp[Ww:].abs().sum()) == 0.0
    outs = []
    for table in (None, xp):
        gxv, gtv = torch.empty(d, N, dtype=torch.float64).cuda(), torch.empty(N, dtype=torch.float64).cuda()
        rec = torch.empty(KN.disc_act_rows(Ww, q), KN.disc_act_cols(L * N), dtype=torch.float64).cuda()
        v, vt = KN.disc_fwd(xT, tc, blob, Ww, q, gxv=gxv, gtv=gtv, ngrad=N, act=rec, max_blocks=7, xproj=table)
        vbar = torch.randn(L, N, dtype=torch.float64, generator=torch.Generator().manual_seed(73)).cuda()
        g = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar, Ww, q, act=rec))
        v3, vt3 = KN.disc_fwd(xT, tc, blob, Ww, q, xproj=table)                  # (no record, no gradient, static split)
        _close(v3, v, 1e-14, 'v without the extras'); _close(vt3, vt, 1e-14, 'dv/dt without the extras')
        outs.append((v, vt, gxv, gtv, g))
    Xd = X.double().requires_grad_(True)
    v_ref = R.v_net(phi, cfg, Xd)
    gX = torch.autograd.grad(v_ref.sum(), Xd)[0]
    _close(outs[1][0].t(), v_ref, 1e-12, 'v'); _close(outs[1][1].t(), gX[:, :, 0], 1e-11, 'dv/dt')
    _close(outs[1][2].t(), gX[:, 0, 1:], 1e-11, 'fused nabla_x v'); _close(outs[1][3], gX[:, 0, 0], 1e-11, 'fused dv/dt')
    for a, b_, what in zip(outs[0], outs[1], ('v', 'vt', 'gxv', 'gtv', 'gradient')):
        _close(b_, a, 1e-12, what + ' hoisted vs plain')
    with pytest.raises(XnwanError):                                              # point mode has no per-path table
        KN.disc_fwd(xT, None, blob, Ww, q, tpp=torch.rand(N, dtype=torch.float64).cuda(), xproj=xp)


def test_disc_forward_ticket_queue_matches_static_split_over_many_launches():
    """more tiles than waves: the tiles after a wave's first one come from ticket counters (k_disc_fwd DYN) that the last wave
    of a launch zeroes again.  Which wave computes a tile cannot change its result: every launch -- more of them than there are
    queue slots, over several grid caps, with and without the record and the fused gradient -- must be BIT-identical to a launch
    with one wave per tile (static split)."""
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d = 1000, 7, 20                                    # 438 tiles (the last one ragged)
    _, phi = _params(d, 8, 51)
    x, t, _ = _sample(N, L, d, 52)
    xT, tc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER)
    gx0, gt0 = torch.empty(d, N, dtype=torch.float64).cuda(), torch.empty(N, dtype=torch.float64).cuda()
    v0, vt0 = KN.disc_fwd(xT, tc, blob, W, Q, gxv=gx0, gtv=gt0, ngrad=N)          # 110 blocks: one tile per wave
    rec0 = torch.zeros(KN.disc_act_rows(W, Q), KN.disc_act_cols(L * N), dtype=torch.float64).cuda()
    KN.disc_fwd(xT, tc, blob, W, Q, act=rec0)
    for k in range(150):
        cap = (1, 2, 3, 7, 33, 40, 64, 100)[k % 8]
        gx, gt = torch.full_like(gx0, float('nan')), torch.full_like(gt0, float('nan'))
        v, vt = KN.disc_fwd(xT, tc, blob, W, Q, gxv=gx, gtv=gt, ngrad=N, max_blocks=cap)
        assert torch.equal(v, v0) and torch.equal(vt, vt0) and torch.equal(gx, gx0) and torch.equal(gt, gt0), (k, cap)
        if k % 10 == 0:
            rec = torch.zeros_like(rec0)
            v, vt = KN.disc_fwd(xT, tc, blob, W, Q, act=rec, max_blocks=cap)
            assert torch.equal(v, v0) and torch.equal(vt, vt0) and torch.equal(rec, rec0), (k, cap)


@pytest.mark.parametrize('N,d', [(37, 5), (64, 20), (20, 70)])
def test_disc_input_gradient(N, d):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    _, phi = _params(d, 8, 31)
    x, t, X = _sample(N, 4, d, 32)
    X0 = X[:, 0, :].double().requires_grad_(True)
    g = torch.autograd.grad(R.v_net(phi, _cfg(), X0).sum(), X0)[0]
    gxv, gtv = KN.disc_gradx(x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER), W, Q)
    _close(gxv.t(), g[:, 1:], 1e-11, 'nabla_x v')
    _close(gtv, g[:, 0], 1e-11, 'dv/dt')


@pytest.mark.parametrize('N,L,d', CASES + [(50, 3, 70), (300, 5, 6), (1100, 32, 6)])     # (last: more 64-point groups than blocks)
def test_disc_backward(N, L, d):
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    _, phi = _params(d, 8, 41)
    ph = {k: v.clone().requires_grad_(True) for k, v in phi.items()}
    x, t, X = _sample(N, L, d, 42)
    vbar = torch.randn(N, L, dtype=torch.float64, generator=torch.Generator().manual_seed(43))
    v_ref = R.v_net(ph, _cfg(), X)
    grads = torch.autograd.grad((v_ref * vbar).sum(), [ph[k] for k in V_ORDER])
    slab = KN.disc_bwd(x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER), vbar.t().contiguous().cuda(), W, Q)
    got = KN.slab_sum(slab).cpu()
    ref = torch.cat([g_.reshape(-1) for g_ in grads])
    off = 0
    for k, g_ in zip(V_ORDER, grads):
        n = g_.numel()
        _close(got[off:off + n], g_.reshape(-1), 1e-10 * max(1.0, float(ref.abs().max()) / max(float(g_.abs().max()), 1e-300)), 'grad ' + k)
        off += n


@pytest.mark.parametrize('q', [0, 1, 4, 12])
@pytest.mark.parametrize('N,L,d', [(37, 4, 5), (50, 3, 70)])
def test_disc_other_depths(N, L, d, q):
    """v_layers other than the YAML's 9: the forward takes the depth at run time, the reverse kernels run from the record
    (the C ABI refuses the recomputing form, the host wrapper stores the record first)"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN, _lib
    cfg = dict(_cfg(), v_layers=q)
    torch.manual_seed(61)
    _, phi = R.init_parameters(cfg, _setup(d, 2))
    for p_ in phi.values():
        if p_.dim() == 1:
            p_.copy_(0.3 * torch.randn_like(p_))
    ph = {k: v.clone().requires_grad_(True) for k, v in phi.items()}
    x, t, X = _sample(N, L, d, 62)
    vbar = torch.randn(N, L, dtype=F64, generator=torch.Generator().manual_seed(63))
    Xd = X.double().requires_grad_(True)
    v_ref = R.v_net(ph, cfg, Xd)
    used = [k for k in V_ORDER if q > 0 or not k.startswith('Vh')]
    grads = dict(zip(used, torch.autograd.grad((v_ref * vbar).sum(), [ph[k] for k in used], retain_graph=True)))
    gX = torch.autograd.grad(v_ref.sum(), Xd)[0]
    xT, tc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER)
    v, vt = KN.disc_fwd(xT, tc, blob, W, q)
    _close(v.t(), v_ref, 1e-12, 'v'); _close(vt.t(), gX[:, :, 0], 1e-11, 'dv/dt')
    got = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar.t().contiguous().cuda(), W, q)).cpu()
    ref = torch.cat([(grads[k] if k in grads else torch.zeros_like(phi[k])).reshape(-1) for k in V_ORDER])
    _close(got, ref, 1e-10, 'phi gradient')
    w0 = torch.randn(N, dtype=F64, generator=torch.Generator().manual_seed(64))
    gxv, gtv = KN.disc_gradx(xT, tc, blob, W, q, vbar=w0.cuda())
    _close(gxv.t(), gX[:, 0, 1:] * w0[:, None], 1e-11, 'nabla_x v'); _close(gtv, gX[:, 0, 0] * w0, 1e-11, 'dv/dt at t0')
    if q != 9:   # the C entry point itself: no silent fallback
        slab = torch.empty(KN.disc_bwd_slabs(N, L), blob.numel(), dtype=F64, device='cuda')
        rc = _lib.lib.xw_disc_bwd(KN._p(xT), KN._p(tc), None, KN._p(blob), None, N, L, d, W, q, None, KN._p(slab), None)
        assert rc == -1


@pytest.mark.parametrize('N,L,d', [(37, 4, 5), (64, 3, 20), (50, 3, 70)])
def test_disc_backward_from_stored_activations(N, L, d):
    """xw_disc_fwd can leave the layer inputs of every point behind; xw_disc_bwd given that record skips its forward
    recompute and must return the same parameter gradient"""
    from xnode_wan_pde_solver_amd import kernels as KN
    _, phi = _params(d, 8, 51)
    x, t, _ = _sample(N, L, d, 52)
    g = torch.Generator().manual_seed(53)
    vbar = torch.randn(L, N, dtype=F64, generator=g).cuda()
    xT, tc, blob = x.double().t().contiguous().cuda(), t.double().cuda(), _blob(phi, V_ORDER)
    rows = KN.disc_act_rows(W, Q)
    assert rows == (Q + 1) * W
    act = torch.full((rows, KN.disc_act_cols(L * N)), float('nan'), dtype=F64, device='cuda')
    v0, vt0 = KN.disc_fwd(xT, tc, blob, W, Q)
    v1, vt1 = KN.disc_fwd(xT, tc, blob, W, Q, act=act)
    assert torch.equal(v0, v1) and torch.equal(vt0, vt1)
    rec = KN.disc_act_view(act, W, Q)                           # [tiles, (Q+1) W rows, 16 points]
    assert torch.isfinite(rec).all() and bool((rec[:, :Q * W, :] >= 0).all())    # relu outputs, every slot written
    assert bool((rec[:, Q * W:, :].abs() <= 1).all())                                 # tanh(a_q)
    s0 = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar, W, Q))
    s1 = KN.slab_sum(KN.disc_bwd(xT, tc, blob, vbar, W, Q, act=act))
    _close(s1, s0, 1e-12, 'phi gradient from the stored record')


@pytest.mark.parametrize('N,L,d', [(1, 2, 1), (5, 3, 2), (19, 3, 126)])
def test_smallest_and_widest_inputs(N, L, d):
    """one path, two time points, a 1-D problem; and the widest input the test network's kernels take (d + 2 = 128)"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    theta, phi = _params(d, 8, 71)
    th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
    ph = {k: v.clone().requires_grad_(True) for k, v in phi.items()}
    x, t, X = _sample(N, L, d, 72)
    g = torch.Generator().manual_seed(73)
    start = torch.randn(N, dtype=F64, generator=g).requires_grad_(True)
    ubar, vbar = torch.randn(N, L, dtype=F64, generator=g), torch.randn(N, L, dtype=F64, generator=g)
    x64 = x.double().requires_grad_(True)
    Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
    u_ref = R.u_net(th, _cfg(), Xd, start)
    gu = torch.autograd.grad((u_ref * ubar).sum(), [x64, start] + [th[k] for k in U_ORDER])
    v_ref = R.v_net(ph, _cfg(), Xd)
    gv = torch.autograd.grad((v_ref * vbar).sum(), [ph[k] for k in V_ORDER], retain_graph=True)
    gXv = torch.autograd.grad(v_ref.sum(), x64)[0]
    xT, tc, sc = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda()
    bu, bv = _blob(theta, U_ORDER), _blob(phi, V_ORDER)
    act = torch.empty(L - 1, KN.ode_act_rows(1, H, K, 8), KN.ode_act_cols(N), dtype=F64, device='cuda')
    ub = ubar.t().contiguous().cuda()
    for a_ in (None, act):
        u, Y = torch.empty(L, N, dtype=F64, device='cuda'), torch.empty(L, H, N, dtype=F64, device='cuda')
        job = dict(xT=xT, start=sc, u=u, Y=Y, act=a_)
        KN.ode_fwd_multi([job], tc, bu, 1, H, K, 8)
        _close(u.t(), u_ref, 1e-12, 'u')
        gx, gs = torch.empty(d, N, dtype=F64, device='cuda'), torch.empty(N, dtype=F64, device='cuda')
        slab = torch.empty(KN.ode_bwd_slabs(N), bu.numel(), dtype=F64, device='cuda')
        KN.ode_bwd_multi([dict(job, ubar=ub, gx=gx, gs=gs, gslab=slab)], tc, bu, 1, H, K, 8, want_x=True, want_params=True)
        _close(gx.t(), gu[0], 1e-10, 'gx'); _close(gs, gu[1], 1e-10, 'gs')
        _close(KN.slab_sum(slab), torch.cat([g_.reshape(-1) for g_ in gu[2:]]), 1e-10, 'theta gradient')
    gxv = torch.empty(d, N, dtype=F64, device='cuda'); gtv = torch.empty(N, dtype=F64, device='cuda')
    rec = torch.empty(KN.disc_act_rows(W, Q), KN.disc_act_cols(L * N), dtype=F64, device='cuda')
    v, vt = KN.disc_fwd(xT, tc, bv, W, Q, gxv=gxv, gtv=gtv, ngrad=N, act=rec)
    _close(v.t(), v_ref, 1e-12, 'v')
    # (x enters every time slice of Xd: the fused gradient is the one of slice 0 only)
    X0 = torch.cat((t.double()[:1].expand(N, 1), x.double()), 1).requires_grad_(True)
    g0 = torch.autograd.grad(R.v_net(phi, _cfg(), X0).sum(), X0)[0]
    _close(gxv.t(), g0[:, 1:], 1e-11, 'nabla_x v at t0'); _close(gtv, g0[:, 0], 1e-11, 'dv/dt at t0')
    ref = torch.cat([g_.reshape(-1) for g_ in gv])
    for a_ in (None, rec):
        _close(KN.slab_sum(KN.disc_bwd(xT, tc, bv, vbar.t().contiguous().cuda(), W, Q, act=a_)), ref, 1e-10, 'phi gradient')


def test_generator_cotangents_split_and_merged_forms():
    """ubarA (pollution + initial penalty), ubarB (= dI/du) and the merged form A + (2/I) B against the closed formulas
    (src/loss.py:55,64,70,79,93)"""
    from xnode_wan_pde_solver_amd import kernels as KN
    torch.manual_seed(5)
    L, N, Vol, Nglob, alpha, kappa = 6, 37, 2.5, 74.0, 1e3, -0.7
    u, v = torch.randn(L, N, dtype=F64), torch.randn(L, N, dtype=F64)
    w, h = torch.rand(N, dtype=F64), torch.randn(N, dtype=F64)
    cN, cNL = Vol / Nglob, Vol / Nglob / L
    refB = cNL * (2 * kappa * u) * v * w
    refB[L - 1] += cN * v[L - 1]
    refA = torch.ones(L, N, dtype=F64)
    refA[0] += alpha * 2 * (u[0] - h) / Nglob
    dev = lambda x: x.cuda()   # noqa: E731
    A, B, Mg = (torch.empty(L, N, dtype=F64, device='cuda') for _ in range(3))
    KN.gen_cotangents(dev(u), dev(v), dev(w), dev(h), Vol, Nglob, alpha, A, B, ckappa=kappa)
    scal = torch.zeros(16, dtype=F64, device='cuda')
    scal[0] = -1.7
    KN.gen_cotangents(dev(u), dev(v), dev(w), dev(h), Vol, Nglob, alpha, Mg, None, ckappa=kappa, scal=scal)
    np.testing.assert_allclose(A.cpu().numpy(), refA.numpy(), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(B.cpu().numpy(), refB.numpy(), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Mg.cpu().numpy(), (refA + (2 / -1.7) * refB).numpy(), rtol=1e-13, atol=1e-13)


def test_adam_matches_torch_formula():
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    g = torch.Generator().manual_seed(51)
    P = 1651
    p0 = torch.randn(P, dtype=torch.float64, generator=g)
    p = {'p': p0.clone()}
    state = {}
    pc, m, v = p0.clone().cuda(), torch.zeros(P, dtype=torch.float64).cuda(), torch.zeros(P, dtype=torch.float64).cuda()
    step = torch.zeros(1, dtype=torch.int64).cuda()
    scal = torch.zeros(16, dtype=torch.float64)
    for it in range(3):
        slabs = torch.randn(37, P, dtype=torch.float64, generator=g)
        slabsB = torch.randn(5, P, dtype=torch.float64, generator=g)
        extra = torch.randn(P, dtype=torch.float64, generator=g)
        scal[0] = 0.7 + it
        p = R.adam_update(p, {'p': slabs.sum(0) + extra + (2.0 / scal[0]) * slabsB.sum(0)}, state, 0.015)
        KN.adam(pc, slabs.cuda(), m, v, step, 0.015, gextraA=extra.cuda(), gslabB=slabsB.cuda(), scal=scal.cuda())
    assert int(step.item()) == 3
    _close(pc, p['p'], 1e-13, 'adam')


def test_dims_outside_the_compiled_set_fail_loudly():
    from xnode_wan_pde_solver_amd import kernels as KN
    from xnode_wan_pde_solver_amd._lib import XnwanError
    x = torch.zeros(3, 16, dtype=torch.float64).cuda()
    t = torch.linspace(0, 1, 4, dtype=torch.float64).cuda()
    # (widths between the MFMA containers -- (24, 12) here -- are served at their own widths by the generic path; beyond ITS
    #  limits, u_hidden_dim 64 / u_hidden_hidden_dim 16 / v_hidden_dim 128, the refusal is loud)
    KN.ode_fwd(x, t, torch.zeros(16, dtype=torch.float64).cuda(), torch.zeros(KN.theta_size(3, 24, 12), dtype=torch.float64).cuda(), 1, 24, 12, 8)
    with pytest.raises(XnwanError):
        KN.ode_fwd(x, t, torch.zeros(16, dtype=torch.float64).cuda(),
                   torch.zeros(KN.theta_size(3, 65, 12), dtype=torch.float64).cuda(), 1, 65, 12, 8)
    with pytest.raises(XnwanError):
        KN.ode_container(20, 17)
    with pytest.raises(XnwanError):
        KN.disc_container(129)
    with pytest.raises(XnwanError):
        KN.ode_fwd(x.cpu(), t, torch.zeros(16, dtype=torch.float64).cuda(),
                   torch.zeros(KN.theta_size(3, 20, 10), dtype=torch.float64).cuda(), 1, 20, 10, 8)


def test_empty_and_malformed_inputs_are_refused_by_the_c_abi():
    """the library never launches on an empty group or with missing / inconsistent buffers: negative XW_E_* status codes"""
    import ctypes
    from xnode_wan_pde_solver_amd import kernels as KN, _lib
    from xnode_wan_pde_solver_amd._lib import XnwanError, lib
    d, N, L = 3, 16, 4
    th = torch.zeros(KN.theta_size(d, H, K), dtype=F64).cuda()
    ph = torch.zeros(KN.phi_size(d, W), dtype=F64).cuda()
    x, t, s = torch.zeros(d, N, dtype=F64).cuda(), torch.linspace(0, 1, L, dtype=F64).cuda(), torch.zeros(N, dtype=F64).cuda()
    u = torch.empty(L, N, dtype=F64).cuda()
    p = lambda a: ctypes.c_void_p(a.data_ptr())   # noqa: E731
    null = ctypes.c_void_p(0)
    # N = 0, L = 0, missing pointers, unknown method, mode 0, x_cot_ones without both outputs
    assert lib.xw_ode_fwd(p(x), p(t), p(s), p(th), 1, 0, L, d, H, K, 8, p(u), null, null) < 0
    assert lib.xw_ode_fwd(p(x), p(t), p(s), p(th), 1, N, 0, d, H, K, 8, p(u), null, null) < 0
    assert lib.xw_ode_fwd(null, p(t), p(s), p(th), 1, N, L, d, H, K, 8, p(u), null, null) < 0
    assert lib.xw_ode_fwd(p(x), p(t), p(s), p(th), 7, N, L, d, H, K, 8, p(u), null, null) < 0
    Y = torch.empty(L, H, N, dtype=F64).cuda()
    gx, gs = torch.empty(d, N, dtype=F64).cuda(), torch.empty(N, dtype=F64).cuda()
    assert lib.xw_ode_bwd(p(x), p(t), p(s), p(th), p(Y), null, 1, N, L, d, H, K, 8, 0, p(gx), p(gs), null, null) < 0
    assert lib.xw_ode_bwd(p(x), p(t), p(s), p(th), p(Y), null, 1, N, L, d, H, K, 8, 5, p(gx), p(gs), null, null) < 0
    assert lib.xw_ode_bwd(p(x), p(t), p(s), p(th), p(Y), null, 1, N, L, d, H, K, 8, 2, null, null, null, null) < 0   # no slabs
    v = torch.empty(L, N, dtype=F64).cuda()
    assert lib.xw_disc_fwd(p(x), p(t), null, p(ph), 0, L, d, W, Q, p(v), null, null, null, 0, 0, null, null) < 0
    assert lib.xw_disc_fwd(p(x), null, null, p(ph), N, L, d, W, Q, p(v), null, null, null, 0, 0, null, null) < 0       # no times
    assert lib.xw_disc_fwd(p(x), p(t), p(s), p(ph), N, L, d, W, Q, p(v), null, null, null, 0, 0, null, null) < 0       # tpp with L > 1
    assert lib.xw_disc_fwd(p(x), p(t), null, p(ph), N, L, d, 129, Q, p(v), null, null, null, 0, 0, null, null) == -1   # width beyond the generic path's 128: XW_E_DIMS
    assert lib.xw_ode_fwd(p(x), p(t), p(s), p(th), 1, N, L, d, 65, K, 8, p(u), null, null) == -1                          # u_hidden_dim beyond its 64
    # and the Python layer turns shape / dtype mismatches into XnwanError before anything reaches the device
    with pytest.raises(XnwanError):
        KN.disc_fwd(x, t, ph[:-1].contiguous(), W, Q)
    with pytest.raises(XnwanError):
        KN.ode_fwd(x.float(), t, s, th, 1, H, K, 8)
    with pytest.raises(XnwanError):
        KN.ode_bwd_multi([dict(xT=x, start=s, Y=Y, ubar=None, gx=gx, gs=gs)], t, th, 1, H, K, 8, want_x=True, want_params=False,
                         x_cot_ones=True)


# ---- widths other than the YAML's (src/model.py:62-85,130-138 accept any u_hidden_dim / u_hidden_hidden_dim) ----------------
@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('Hh,Kk,m', [(32, 12, 8), (32, 12, 3), (32, 12, 1), (32, 12, 10), (20, 10, 10),
                                     (48, 16, 8), (64, 16, 3), (33, 9, 1), (24, 13, 10),          # ((48, 16), (33, 9), (24, 13) as they are: the generic path)
                                     (64, 16, 8), (64, 16, 9), (64, 16, 1), (64, 16, 10),        # ((64, 16): the wide MFMA container, round 6; depth 10: two mask words)
                                     (64, 16, 11), (20, 10, 12), (24, 13, 17)])                  # (deeper than the containers: the generic path, u_layers <= 32)
def test_ode_kernels_at_the_wide_instantiation(Hh, Kk, m, solver):
    """the (32, 12) stepper object: H a multiple of 16 (the time row of [y ; t] is a tile of its own), K = 12 (no padding
    row inside the 4-row blocks) -- forward 1e-12, sweep (x, start, every weight gradient) 1e-10 against the oracle,
    with and without the activation store (duo sweep / single-wave sweep)"""
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import kernels as KN
    N, L, d = 37, 6, 5
    cfg = dict(_cfg(m, solver), u_hidden_dim=Hh, u_hidden_hidden_dim=Kk)
    torch.manual_seed(31)
    theta, _ = R.init_parameters(cfg, _setup(d, 2))
    for p in theta.values():
        if p.dim() == 1:
            p.copy_(0.3 * torch.randn(p.shape, dtype=torch.float64))
    th = {k: v.clone().requires_grad_(True) for k, v in theta.items()}
    x, t, X = _sample(N, L, d, 32)
    g = torch.Generator().manual_seed(33)
    start = torch.randn(N, dtype=torch.float64, generator=g).requires_grad_(True)
    ubar = torch.randn(N, L, dtype=torch.float64, generator=g)
    x64 = x.double().requires_grad_(True)
    Xd = torch.cat((t.double().view(1, L, 1).expand(N, L, 1), x64.view(N, 1, d).expand(N, L, d)), 2)
    u_ref = R.u_net(th, cfg, Xd, start)
    order = [k for k in U_ORDER if k in theta]
    grads = torch.autograd.grad((u_ref * ubar).sum(), [x64, start] + [th[k] for k in order], allow_unused=True)
    grads = [g_ if g_ is not None else torch.zeros_like(p_) for g_, p_ in zip(grads, [x64, start] + [th[k] for k in order])]
    blob = torch.cat([(theta[k] if k in theta else torch.zeros(Kk * Kk if k == 'Wh' else Kk, dtype=torch.float64)).reshape(-1)
                      for k in U_ORDER]).cuda()
    xT, tc, sc = x.double().t().contiguous().cuda(), t.double().cuda(), start.detach().cuda()
    mid = KN.method_id(solver)
    u, Y = KN.ode_fwd(xT, tc, sc, blob, mid, Hh, Kk, m)
    _close(u.t(), u_ref.detach(), 1e-12, 'u')
    rows = KN.ode_act_rows(mid, Hh, Kk, m)
    for with_act in ([False, True] if rows else [False]):
        job = dict(xT=xT, start=sc, u=torch.empty_like(u), Y=torch.empty_like(Y))
        if with_act:
            job['act'] = torch.empty(L - 1, rows, KN.ode_act_cols(N), dtype=torch.float64, device='cuda')
        KN.ode_fwd_multi([job], tc, blob, mid, Hh, Kk, m)
        gx, gs = torch.empty(d, N, dtype=torch.float64, device='cuda'), torch.empty(N, dtype=torch.float64, device='cuda')
        slab = torch.empty(KN.ode_bwd_slabs(N), blob.numel(), dtype=torch.float64, device='cuda')
        KN.ode_bwd_multi([dict(job, ubar=ubar.t().contiguous().cuda(), gx=gx, gs=gs, gslab=slab)], tc, blob, mid, Hh, Kk, m,
                         want_x=True, want_params=True)
        _close(gx.t(), grads[0], 1e-10, 'gx act=%s' % with_act)
        _close(gs, grads[1], 1e-10, 'gs act=%s' % with_act)
        flat, off = KN.slab_sum(slab).cpu(), 0
        for k in U_ORDER:
            n = theta[k].numel() if k in theta else (Kk * Kk if k == 'Wh' else Kk)
            if k in theta:
                _close(flat[off:off + n].view(theta[k].shape), grads[2 + order.index(k)], 1e-10, 'grad %s act=%s' % (k, with_act))
            off += n


@pytest.mark.parametrize('solver', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('form', ['adjoint', 'ones', 'store', 'residual', 'smallest'])
def test_ode_launch_forms_at_the_widest_container(monkeypatch, form, solver):
    """the (64, 16) container (round 6: the field on 16x16x4 matrix instructions, one wave per tile) under every launch form the
    engine uses -- the continuous adjoint, the all-ones x cotangent of two jobs in one launch, the activation store and its
    x-only form, cotangents formed inside the sweep: the tests of the (20, 10) container above, run at these widths"""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, 'H', 64)
    monkeypatch.setattr(mod, 'K', 16)
    if form == 'adjoint':
        test_ode_backward_continuous_adjoint(37, 7, 5, solver)
    elif form == 'ones':
        test_ode_backward_pollution_and_x_sweep_in_one(solver)
    elif form == 'store':
        test_ode_backward_from_stored_activations(solver)
    elif form == 'smallest':
        if solver == 'midpoint':              # (one path and two time points; 19 paths at the widest input, d = 126)
            test_smallest_and_widest_inputs(1, 2, 1)
            test_smallest_and_widest_inputs(19, 3, 126)
    else:
        test_ode_sweep_with_residual_cotangents(solver)


@pytest.mark.parametrize('N,d', [(1, 1), (37, 2), (300, 5), (4096, 20), (1000, 100)])
def test_cube_weight_kernel_is_the_tensor_formulation_bit_for_bit(N, d):
    """xw_cube_weight against Hypercube.func_w_grad (itself checked against autograd through func_w, the reference's
    src/dataset.py:278-282 + src/loss.py:51-63) on the device: w, the gradient and the transposed points, including points ON
    faces, exact ties between the two distances (x_i = 0 on a symmetric cube), between coordinates, and the autograd result."""
    from xnode_wan_pde_solver_amd import sampling
    from xnode_wan_pde_solver_amd import kernels as KN
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(N * 131 + d)
    for top_bot in ((-1.0, 1.0), (-0.3, 2.5)):
        dom = sampling.Hypercube(top_bot, d, 0.0, 1.0, 4)
        x = (torch.rand(N, d, generator=g) * (top_bot[1] - top_bot[0]) + top_bot[0])
        if N > 8:
            x[0, 0] = top_bot[1]                       # on the top face
            x[1, d - 1] = top_bot[0]                   # on the bottom face
            x[2] = 0.5 * (top_bot[0] + top_bot[1])     # centre: every coordinate ties, and top ties with bottom
            x[3, :] = x[3, 0]                          # all coordinates equal
            x[4, d // 2] = x[4, 0]
        x = x.to(dev)
        X = sampling._paths(torch.zeros(1, device=dev), x)                 # [N, 1, 1+d]
        w_ref, g_ref = dom.func_w_grad(X)
        Xa = X.clone().requires_grad_(True)
        g_auto = torch.autograd.grad(dom.func_w(Xa).sum(), Xa)[0]
        assert torch.equal(g_auto, g_ref)
        w = torch.empty(N, dtype=torch.float64, device=dev)
        w0 = torch.empty_like(w)
        gwT = torch.empty(d, N, dtype=torch.float64, device=dev)
        xT = torch.empty(d, N, dtype=torch.float64, device=dev)
        KN.cube_weight(x, dom.top, dom.bot, w, gwT, w0=w0, xT=xT)
        assert torch.equal(w, w_ref[:, 0].double()) and torch.equal(w0, w)
        assert torch.equal(gwT, g_ref[:, 0, 1:].double().t())
        assert torch.equal(xT, x.double().t())
        KN.cube_weight(x, dom.top, dom.bot, w, gwT)                        # the optional outputs left out
        assert torch.equal(gwT, g_ref[:, 0, 1:].double().t())
