"""Parity at the BASELINE.json sizes (where the oracle would take minutes): size-independent properties of the kernels.

Sizes: configs[1] (headline: d=20, N=4096 paths, L=32 times), the per-GPU shares of configs[2] (d=50, N=2048, L=64) and
configs[3] (d=100, N=8192, L=32), and configs[4] (Ex4_3 on the time-varying balls, d=10, N_r=N_b=8192, N_t=20) whole.
Properties: linearity of the reverse sweeps / the test-network backward in their cotangent, agreement of the fused and
the separate forms (pollution sweep + nabla_x u, stored activations vs recompute, fused vs stand-alone input gradient),
equivariance under a permutation of the paths, the reductions against a second formulation in torch on the same device,
bit-reproducibility of a whole sub-step.  Every call goes through the C ABI (kernels.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F64 = torch.float64
H, K, M, W, Q = 20, 10, 8, 50, 9
SIZES = [(20, 4096, 32), (50, 2048, 64), (100, 8192, 32),      # configs[1]; per-GPU shares of configs[2] and configs[3]
         (50, 16384, 64), (100, 65536, 32)]                      # configs[2] and configs[3] WHOLE (their stated global batch) on one GPU


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


def _setup(d, N, L, seed):
    from xnode_wan_pde_solver_amd import kernels as KN, _lib
    g = torch.Generator().manual_seed(seed)
    dev = torch.device('cuda')
    th = (0.3 * torch.randn(_lib.lib.xw_theta_size(d, H, K), generator=g, dtype=F64)).to(dev)
    ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=F64)).to(dev)
    xT = (torch.rand(d, N, generator=g, dtype=F64) * 2 - 1).to(dev)
    t = torch.sort(torch.rand(L, generator=g, dtype=F64)).values
    t[0], t[-1] = 0.0, 1.0
    start = torch.randn(N, generator=g, dtype=F64).to(dev)
    return KN, g, dev, th, ph, xT, t.to(dev), start


@pytest.mark.parametrize('narrow', [False, True])
@pytest.mark.parametrize('d,N,L', SIZES)
def test_stepper_sweeps_at_full_size(d, N, L, narrow):
    """narrow: the same properties for the narrow-tile kernels (four waves of 4 paths per tile, csrc/xw_ode_n4.h) -- the
    forward pass that fills the store and every sweep that reads it"""
    if narrow and N > 16384:
        pytest.skip('narrow tiles are a small-launch layout; exercised up to 16384 paths')
    KN, g, dev, th, ph, xT, t, start = _setup(d, N, L, 1)
    Mth = (1, H, K, M)                                             # midpoint
    rows = KN.ode_act_rows(1, H, K, M)
    u, Y = torch.empty(L, N, dtype=F64, device=dev), torch.empty(L, H, N, dtype=F64, device=dev)
    act = torch.empty(L - 1, rows, KN.ode_act_cols(N), dtype=F64, device=dev)
    job = dict(xT=xT, start=start, u=u, Y=Y, act=act)
    KN.ode_fwd_multi([job], t, th, *Mth, narrow=narrow)
    assert torch.isfinite(u).all() and torch.isfinite(act).all()
    u_plain, Y_plain = KN.ode_fwd(xT, t, start, th, 1, H, K, M)
    if narrow:                                                     # (another summation order inside a layer)
        assert _rel(u, u_plain) < 1e-12 and _rel(Y, Y_plain) < 1e-12
    else:
        assert torch.equal(u, u_plain) and torch.equal(Y, Y_plain)     # storing the activations does not change the forward

    def sweep(ubar, with_act=True, x_ones=False):
        gx, gs = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
        slab = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=F64, device=dev)
        j = dict(job, ubar=ubar, gx=gx, gs=gs, gslab=slab)
        if not with_act:
            j['act'] = None
        KN.ode_bwd_multi([j], t, th, *Mth, want_x=True, want_params=True, x_cot_ones=x_ones, narrow=narrow and with_act)
        return gx, gs, KN.slab_sum(slab)

    u1 = torch.randn(L, N, generator=g, dtype=F64).to(dev)
    u2 = torch.randn(L, N, generator=g, dtype=F64).to(dev)
    a, b, c = sweep(u1), sweep(u2), sweep(u1 + 2.0 * u2)
    for x1, x2, x12, what in zip(a, b, c, ('gx', 'gs', 'theta gradient')):           # linearity in the cotangent
        assert _rel(x12, x1 + 2.0 * x2) < 1e-11, what
    r = sweep(u1, with_act=False)                                                     # stored activations == recompute
    for x1, x2, what in zip(a, r, ('gx', 'gs', 'theta gradient')):
        assert _rel(x1, x2) < 1e-11, what
    ones = torch.ones(L, N, dtype=F64, device=dev)                                    # pollution sweep + nabla_x u in one
    poll = ones.clone()
    poll[0] += 30.0 * torch.randn(N, generator=g, dtype=F64).to(dev)
    gx1, gs1, _ = sweep(ones)
    gx7, gs7, th7 = sweep(poll, x_ones=True)
    _, _, thp = sweep(poll)
    assert _rel(gx7, gx1) < 1e-12 and _rel(gs7, gs1) < 1e-10 and torch.equal(th7, thp)
    if narrow:                                                     # the sweep WITHOUT weight gradients is a kernel of its own
        gxo, gso = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
        KN.ode_bwd_multi([dict(job, ubar=u1, gx=gxo, gs=gso)], t, th, *Mth, want_x=True, want_params=False, narrow=True)
        assert _rel(gxo, a[0]) < 1e-12 and _rel(gso, a[1]) < 1e-12
    # equivariance: permuting the paths permutes the per-path outputs and leaves the parameter gradient alone
    perm = torch.randperm(N, generator=g).to(dev)
    jobp = dict(xT=xT[:, perm].contiguous(), start=start[perm].contiguous(), u=torch.empty_like(u), Y=torch.empty_like(Y),
                act=torch.empty_like(act))
    KN.ode_fwd_multi([jobp], t, th, *Mth, narrow=narrow)
    assert torch.equal(jobp['u'], u[:, perm])
    gxp, gsp = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
    slabp = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=F64, device=dev)
    KN.ode_bwd_multi([dict(jobp, ubar=u1[:, perm].contiguous(), gx=gxp, gs=gsp, gslab=slabp)], t, th, *Mth, want_x=True,
                     want_params=True, narrow=narrow)
    assert torch.equal(gxp, a[0][:, perm]) and torch.equal(gsp, a[1][perm])
    assert _rel(KN.slab_sum(slabp), a[2]) < 1e-11


@pytest.mark.parametrize('d,N,L', SIZES[:3])
def test_widest_stepper_container_at_full_size(d, N, L, monkeypatch):
    """the (64, 16) container (round 6: the field on 16x16x4 matrix instructions; the sweep with weight gradients from the
    store is a duo sweep -- chain wave + partner wave --, the recomputing one a single wave): the same properties"""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, 'H', 64)
    monkeypatch.setattr(mod, 'K', 16)
    test_stepper_sweeps_at_full_size(d, N, L, False)


@pytest.mark.parametrize('d,N,L', SIZES)
def test_test_network_at_full_size(d, N, L):
    KN, g, dev, th, ph, xT, t, start = _setup(d, N, L, 2)
    P = N * L
    v, vt = KN.disc_fwd(xT, t, ph, W, Q)
    assert torch.isfinite(v).all() and torch.isfinite(vt).all()
    # the same points in point mode (each (t_l, x_n) as its own point): identical values
    sel = torch.randint(0, P, (4096,), generator=g)
    l_idx, n_idx = (sel // N).to(dev), (sel % N).to(dev)
    vp, vtp = KN.disc_fwd(xT[:, n_idx].contiguous(), None, ph, W, Q, tpp=t[l_idx].contiguous())
    assert _rel(vp.view(-1), v[l_idx, n_idx]) < 1e-13 and _rel(vtp.view(-1), vt[l_idx, n_idx]) < 1e-12
    # d/dt tangent against a centred difference of the kernel itself
    eps = 1e-6
    vplus, _ = KN.disc_fwd(xT, t + eps, ph, W, Q, want_vt=False)
    vminus, _ = KN.disc_fwd(xT, t - eps, ph, W, Q, want_vt=False)
    err = ((vplus - vminus) / (2 * eps) - vt).abs() / float(vt.abs().max())
    # (a piecewise-linear network: the few points with a ReLU kink inside [t - eps, t + eps] are excluded)
    assert float(err.median()) < 1e-8 and float((err > 1e-5).double().mean()) < 2e-3
    # fused input gradient at the first time index == the stand-alone reverse pass
    gxv, gtv = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
    v2, vt2 = KN.disc_fwd(xT, t, ph, W, Q, gxv=gxv, gtv=gtv, ngrad=N)
    gx_ref, gt_ref = KN.disc_gradx(xT, t[:1].contiguous(), ph, W, Q)
    assert torch.equal(v2, v) and _rel(gxv, gx_ref) < 1e-12 and _rel(gtv, gt_ref) < 1e-12 and _rel(gtv, vt[0]) < 1e-12
    # backward: record == recompute, linear in the cotangent
    act = torch.empty(KN.disc_act_rows(W, Q), KN.disc_act_cols(P), dtype=F64, device=dev)
    v3, _ = KN.disc_fwd(xT, t, ph, W, Q, act=act)
    assert torch.equal(v3, v) and torch.isfinite(act).all()      # (P is a multiple of 16: every slot of the record is a point's)
    b1 = torch.randn(L, N, generator=g, dtype=F64).to(dev)
    b2 = torch.randn(L, N, generator=g, dtype=F64).to(dev)
    g1 = KN.slab_sum(KN.disc_bwd(xT, t, ph, b1, W, Q, act=act))
    g2 = KN.slab_sum(KN.disc_bwd(xT, t, ph, b2, W, Q, act=act))
    g12 = KN.slab_sum(KN.disc_bwd(xT, t, ph, b1 - 3.0 * b2, W, Q, act=act))
    g1r = KN.slab_sum(KN.disc_bwd(xT, t, ph, b1, W, Q))
    assert _rel(g12, g1 - 3.0 * g2) < 1e-11 and _rel(g1, g1r) < 1e-11
    # the output bias sees the plain sum of the cotangent (checksum of the whole reduction tree)
    assert abs(float(g1[-1]) - float(b1.sum())) < 1e-9 * max(1.0, float(b1.abs().sum()))


def test_test_network_width_64_at_full_size():
    """the second compiled width at the headline size (d = 20, 4096 paths x 32 times): point mode == path mode, the d/dt
    tangent against a centred difference, fused gradient consistent with the tangent, the reverse from the record linear in
    the cotangent, checksums of the bias gradients (dVo.b = sum of the cotangent; dVh.b, summed on the vector ALU at this
    width, against a centred difference of <vbar, v> in the bias of one unit)"""
    d, N, L, Ww, q = 20, 4096, 32, 64, 9
    from xnode_wan_pde_solver_amd import kernels as KN, _lib
    g = torch.Generator().manual_seed(5)
    dev = torch.device('cuda')
    ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, Ww), generator=g, dtype=F64)).to(dev)
    xT = (torch.rand(d, N, generator=g, dtype=F64) * 2 - 1).to(dev)
    t = torch.sort(torch.rand(L, generator=g, dtype=F64)).values
    t[0], t[-1] = 0.0, 1.0
    t = t.to(dev)
    P = N * L
    gxv, gtv = torch.empty(d, N, dtype=F64, device=dev), torch.empty(N, dtype=F64, device=dev)
    act = torch.empty(KN.disc_act_rows(Ww, q), KN.disc_act_cols(P), dtype=F64, device=dev)
    v, vt = KN.disc_fwd(xT, t, ph, Ww, q, gxv=gxv, gtv=gtv, ngrad=N, act=act, max_blocks=352)      # (ticket queues)
    assert torch.isfinite(v).all() and torch.isfinite(vt).all() and torch.isfinite(act).all()
    assert _rel(gtv, vt[0]) < 1e-12
    sel = torch.randint(0, P, (4096,), generator=g)
    l_idx, n_idx = (sel // N).to(dev), (sel % N).to(dev)
    vp, vtp = KN.disc_fwd(xT[:, n_idx].contiguous(), None, ph, Ww, q, tpp=t[l_idx].contiguous())
    assert _rel(vp.view(-1), v[l_idx, n_idx]) < 1e-13 and _rel(vtp.view(-1), vt[l_idx, n_idx]) < 1e-12
    eps = 1e-6
    vplus, _ = KN.disc_fwd(xT, t + eps, ph, Ww, q, want_vt=False)
    vminus, _ = KN.disc_fwd(xT, t - eps, ph, Ww, q, want_vt=False)
    err = ((vplus - vminus) / (2 * eps) - vt).abs() / float(vt.abs().max())
    assert float(err.median()) < 1e-8 and float((err > 1e-5).double().mean()) < 2e-3
    b1 = torch.randn(L, N, generator=g, dtype=F64).to(dev)
    b2 = torch.randn(L, N, generator=g, dtype=F64).to(dev)
    g1 = KN.slab_sum(KN.disc_bwd(xT, t, ph, b1, Ww, q, act=act))
    g2 = KN.slab_sum(KN.disc_bwd(xT, t, ph, b2, Ww, q, act=act))
    g12 = KN.slab_sum(KN.disc_bwd(xT, t, ph, b1 - 3.0 * b2, Ww, q, act=act))
    assert _rel(g12, g1 - 3.0 * g2) < 1e-11
    assert abs(float(g1[-1]) - float(b1.sum())) < 1e-9 * max(1.0, float(b1.abs().sum()))
    # dVh.b of one hidden unit against a centred difference of sum(v) (all-ones cotangent: no cancellation, so that the
    # handful of points with a ReLU kink inside the step do not dominate the error; blob layout Vin, Vin.b, Vh, Vh.b, Vo, Vo.b)
    ones = torch.ones(L, N, dtype=F64, device=dev)
    g_one = KN.slab_sum(KN.disc_bwd(xT, t, ph, ones, Ww, q, act=act))
    off_vhb = Ww * (d + 1) + Ww + Ww * Ww
    k, h = 37, 1e-7
    php, phm = ph.clone(), ph.clone()
    php[off_vhb + k] += h
    phm[off_vhb + k] -= h
    fp = float(KN.disc_fwd(xT, t, php, Ww, q, want_vt=False)[0].sum())
    fm = float(KN.disc_fwd(xT, t, phm, Ww, q, want_vt=False)[0].sum())
    assert abs((fp - fm) / (2 * h) - float(g_one[off_vhb + k])) < 1e-4 * max(1.0, abs(float(g_one[off_vhb + k]))), ((fp - fm) / (2 * h), float(g_one[off_vhb + k]))


def test_reductions_against_torch_at_full_size():
    """weak_partials / bdry_partials / cotangent kernels at the headline size against the same formulas in torch"""
    d, N, L = 20, 4096, 32
    KN, g, dev, th, ph, xT, t, start = _setup(d, N, L, 3)
    r = lambda *s: torch.randn(*s, generator=g, dtype=F64).to(dev)   # noqa: E731
    u, v, vt, f = r(L, N), r(L, N), r(L, N), r(L, N)
    w, h, gs, w0 = torch.rand(N, generator=g, dtype=F64).to(dev), r(N), r(N), torch.rand(N, generator=g, dtype=F64).to(dev)
    gx, ghT, gxv, gwx0T = r(d, N), r(d, N), r(d, N), r(d, N)
    Vol, kappa, alpha = 3.7, -1.0, 1e4
    scal = torch.zeros(16, dtype=F64, device=dev)
    work = torch.zeros(KN.reduce_work_size(), dtype=F64, device=dev)
    KN.weak_partials(u, v, vt, w, f, h, Vol, float(N), scal, work, ckappa=kappa,
                     contract=dict(gx=gx, gs=gs, ghT=ghT, gxv=gxv, w0=w0, gwx0T=gwx0T))
    cN, cNL = Vol / N, Vol / N / L
    phi, phit = v * w, vt * w
    s31 = ((w0 * gxv + v[0] * gwx0T) * (gx + gs * ghT)).sum(0)
    s3 = kappa * u * u * phi + f * phi
    s3[0] += s31
    I = (cN * (u[-1] * v[-1] - h * v[0])).sum() - cNL * (u * phit - s3).sum()
    np.testing.assert_allclose(float(scal[0]), float(I), rtol=1e-11)
    np.testing.assert_allclose(float(scal[1]), float((v * v).sum()), rtol=1e-12)
    np.testing.assert_allclose(float(scal[2]), float(((u[0] - h) ** 2).sum()), rtol=1e-12)
    ub, gb = r(L, N), r(L, N)
    ubar_b = torch.empty(L, N, dtype=F64, device=dev)
    KN.bdry_partials(ub, gb, alpha, float(N), scal, work, ubar_b=ubar_b)
    np.testing.assert_allclose(float(scal[3]), float(((ub - gb) ** 2).sum()), rtol=1e-12)
    assert _rel(ubar_b, alpha * 2.0 * (ub - gb) / (N * L)) < 1e-13
    vbar = torch.empty(L, N, dtype=F64, device=dev)
    KN.disc_cotangent(u, v, w, f, h, Vol, float(N), scal, vbar, ckappa=kappa)
    dI = cNL * (kappa * u * u + f) * w
    dI[-1] += cN * u[-1]
    dI[0] -= cN * h
    assert _rel(vbar, w - (2.0 / scal[0]) * dI + 2.0 * v / scal[1]) < 1e-12
    # a second launch on the same inputs returns the same bits (deterministic grid sums)
    scal2 = torch.zeros(16, dtype=F64, device=dev)
    KN.weak_partials(u, v, vt, w, f, h, Vol, float(N), scal2, work, ckappa=kappa,
                     contract=dict(gx=gx, gs=gs, ghT=ghT, gxv=gxv, w0=w0, gwx0T=gwx0T))
    assert torch.equal(scal2[:3], scal[:3])
    # the boundary sum of squares formed by the SAME launch (what the generator sub-step does since round 4: xw_weak_partials(ub, gb, Pb)),
    # also with a boundary sample of another size than the interior's
    for ubx, gbx in ((ub, gb), (ub[:7, :1000].contiguous(), gb[:7, :1000].contiguous())):
        scal3 = torch.zeros(16, dtype=F64, device=dev)
        KN.weak_partials(u, v, vt, w, f, h, Vol, float(N), scal3, work, ckappa=kappa,
                         contract=dict(gx=gx, gs=gs, ghT=ghT, gxv=gxv, w0=w0, gwx0T=gwx0T), bdry=dict(ub=ubx, g=gbx))
        assert torch.equal(scal3[:3], scal[:3])
        np.testing.assert_allclose(float(scal3[3]), float(((ubx - gbx) ** 2).sum()), rtol=1e-12)
        assert float(scal3[4:].abs().max()) == 0.0


def test_headline_substeps_are_bit_reproducible():
    """two solvers built from the same seed run g, g, d at the headline size and end with identical parameters"""
    import configs.Ex4_1_funcs as P
    from bench import workload_params
    from src.training import NODE_WAN_solver
    from src.dataset import Comb_loader
    outs = []
    for _ in range(2):
        torch.manual_seed(0)
        np.random.seed(0)
        S = NODE_WAN_solver(workload_params(20, 4096, 4096, 32), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                            torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
        s = S.setup
        domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
        du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, torch.device('cuda'))[0]
        G = S.engine.load_group(du, dv, bd, domain)
        for _ in range(2):
            S.engine.generator_step(G)
            S.engine.generator_step(G)
            S.engine.discriminator_step(G)
        outs.append((S.engine.theta.data.clone(), S.engine.phi.data.clone(), S.engine.scal.clone()))
        assert torch.isfinite(outs[-1][0]).all() and torch.isfinite(outs[-1][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize('name', ['NSphere_TCone', 'NSphere_THourglass'])
def test_config5_time_varying_ball_at_full_size(name, tmp_path, monkeypatch):
    """BASELINE configs[4]: Ex4_3 functions (configs/Ex4_3_funcs.py:6-49 of the reference) on the time-varying balls
    (src/dataset.py:48-229), d = 10, N_r = N_b = 8192, N_t = 20 -- the whole list-domain protocol at full size: the
    groups the engine steps through are the sampler's (every path in exactly one interior group, equal lengths inside a
    group, inside the domain, boundary groups on the boundary), one outer iteration (n1 = 2 generator + n2 = 1
    discriminator sub-iterations, one optimiser step per group) is finite and bit-reproducible from the seed."""
    import configs.Ex4_3_funcs as F
    from src.training import NODE_WAN_solver
    from src.dataset import Comb_loader
    monkeypatch.chdir(tmp_path)
    d, N_r, N_t = 10, 8192, 20
    params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': d, 'N_t': N_t, 'N_r': N_r, 'N_b': N_r, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 1, 'domain': name}
    outs = []
    for rep in range(2):
        torch.manual_seed(11)
        np.random.seed(11)
        S = NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cuda'), './',
                            func_u_sol=F.func_u_sol, p=2)
        assert S.engine.structure.a_identity and S.engine.structure.b_zero and S.engine.structure.c_kappa == -1.0
        if rep == 0:
            rng_t, rng_n = torch.get_rng_state(), np.random.get_state()
            s = S.setup
            domain = S.domain(s['shape_param'], d, s['T0'], s['T'], N_t)
            pts = Comb_loader(N_r, N_r, domain, S.device)
            torch.set_rng_state(rng_t)
            np.random.set_state(rng_n)
            # the sampler's groups
            assert isinstance(pts.interioru, list) and len(pts.interioru) >= 2 and len(pts.boundary) >= 2
            n_paths = 0
            for g_ in pts.interioru:
                assert g_.dtype == F64 and g_.shape[2] == d + 1 and 1 <= g_.shape[1] <= N_t
                assert bool(torch.all(g_[:, 1:, 0] >= g_[:, :-1, 0]))                        # sorted times along a path
                assert bool(torch.all(g_[:, :, 1:] == g_[:, :1, 1:]))                         # vertical paths
                assert float(domain.func_w(g_.detach()).min()) >= -1e-12                      # inside the moving ball
                n_paths += g_.shape[0] if float(g_.detach()[0, 0, 0]) == 0.0 or name == 'NSphere_TCone' else 0
            if name == 'NSphere_TCone':
                assert n_paths == N_r                                                         # every path in exactly one group
            for g_ in pts.boundary:
                assert g_.shape[1] == 1 and float(domain.func_w(g_.detach()).abs().max()) < 1e-9
            n_groups = min(len(pts.interioru), len(pts.boundary))
        losses = S.train()
        assert len(losses) == 2 and all(np.isfinite(losses))
        assert int(S.engine.adam_u['step'].item()) == 2 * n_groups and int(S.engine.adam_v['step'].item()) == n_groups
        assert torch.isfinite(S.engine.theta.data).all() and torch.isfinite(S.engine.phi.data).all()
        outs.append((list(losses), S.engine.theta.data.clone(), S.engine.phi.data.clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
