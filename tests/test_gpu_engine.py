"""End-to-end parity of the HIP engine (through NODE_WAN_solver and the C ABI) against
  (1) the golden vectors recorded from the upstream reference (tests/golden/*.npz), and
  (2) the oracle on the same seeded inputs (other sizes, general coefficients),
plus the trained-error trajectory against the reference's own run.

Tolerances.  v_phi depends only on the float32 sample and the float64 parameters: compared at 1e-10.  Everything
downstream of the user's float32 callables (h, f, g, u_sol -> start values -> u, penalties, losses) is compared at
float32 resolution, because float32 sin/cos/exp differ in the last bit between host CPUs (the fixtures were recorded on
a Xeon, the GPU box has an EPYC; the tight float64 checks of the same kernels are in test_gpu_kernels.py and in the
oracle comparisons below, where both sides tabulate on the same host).  The reference also stores nabla u and nabla phi
in float32 `.grad` tensors while the engine keeps them in float64: I, int, loss_v and the gradients are compared at
1e-5 relative to the largest gradient entry.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import configs.Ex4_1_funcs as P  # noqa: E402

CASES = ['ref_tiny_midpoint', 'ref_tiny_euler', 'ref_tiny_rk4', 'ref_plumb_midpoint', 'ref_d20_small_midpoint',
         'ref_d50_nt64_small_midpoint',       # BASELINE configs[2] family (d = 50, N_t = 64)
         'ref_d100_small_midpoint',           # BASELINE configs[3] family (d = 100, N_t = 32)
         'ref_d20_headline',                  # BASELINE configs[1] AT THE BENCHMARKED SIZE (N_r = N_b = 4096, N_t = 32; slim
                                              # record of the reference's own run: graphs, side streams, ticket queues on)
         'ref_general_d4_midpoint',           # the reference run with GENERAL a_ij, b_i, c(u, t, x) (tests/golden/general_funcs.py)
         # other network shapes, run by the reference itself (round 4): the widest / deepest the engine compiles -- (32, 12) field of
         # depth 10, test network 64 wide --, an odd narrow pair (7, 3) / 11 zero-padded inside the (20, 10) / 50 containers (euler),
         # and a field without hidden layer (u_layers = 1) with a 57-wide, 12-deep test network (rk4)
         'ref_wide_d6_midpoint', 'ref_narrow_d3_euler', 'ref_m1_d4_rk4',
         # round 5: widths beyond the MFMA kernel instantiations, run by the reference itself -- (48, 16) / 100, (64, 16) / 128 (the
         # limits of csrc/xw_generic.hip) and the reference's own field next to a 70-wide test network (MFMA stepper + generic test net)
         'ref_generic_d5_midpoint', 'ref_generic_d3_rk4', 'ref_generic_mixed_d4_euler',
         # round 5: other time intervals and cubes -- [0.25, 1.5] x [-0.5, 1.5]^4 (midpoint), [-1, 0] x [0, 2]^3 (rk4)
         'ref_interval_d4_midpoint', 'ref_interval_d3_rk4',
         # round 5: alpha = 1 -- the interior term's gradient is not hidden behind alpha x penalties in the generator sub-steps
         'ref_alpha1_d4_midpoint', 'ref_alpha1_d3_rk4', 'ref_alpha1_general_d4_euler',
         # round 5: the smallest shapes -- d = 2, two sample times (ONE step), 7 interior / 5 boundary paths; N_t = 3 with rk4
         'ref_min_d2_nt2_midpoint', 'ref_min_d2_nt3_rk4',
         # round 5: the table forms of a -- one constant matrix / a diagonal a(x) -- with the linear reaction c = -0.7 u
         'ref_const_a_d4_midpoint', 'ref_diag_a_d5_rk4',
         # round 5: ONE boundary path; ONE interior path (where the reference's .squeeze() calls also drop the path axis)
         'ref_nb1_d3_midpoint', 'ref_nr1_d3_midpoint']
FUNCS = dict(h=P.func_h, f=P.func_f, g=P.func_g, a=P.func_a, b=P.func_b, c=P.func_c)


def load(golden_dir, case):
    z = np.load(os.path.join(golden_dir, case + '.npz'))
    return z, json.loads(str(z['params_json']))


def make_solver(params, seed, F=P, **kw):
    from src.training import NODE_WAN_solver
    torch.manual_seed(seed)
    np.random.seed(seed)
    return NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cuda'), './',
                           func_u_sol=F.func_u_sol, p=2, **kw)


def _general_funcs(golden_dir, base, with_b):
    """tests/golden/general_funcs.py (the callables the general-coefficient fixtures were recorded with) over `base`'s h, f, g"""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location('general_funcs', os.path.join(golden_dir, 'general_funcs.py'))
    GF = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(GF)
    return types.SimpleNamespace(func_a=GF.func_a, func_b=GF.func_b if with_b else base.func_b, func_c=GF.func_c, func_h=base.func_h,
                                 func_f=base.func_f, func_g=base.func_g, func_u_sol=base.func_u_sol)


def _np(x):
    return x.detach().cpu().double().numpy() if torch.is_tensor(x) else np.asarray(x, dtype=np.float64)


def close(a, b, rtol, atol=0.0, what=''):
    np.testing.assert_allclose(_np(a), _np(b), rtol=rtol, atol=atol, err_msg=what)


F32TOL = 3e-6


def thin(z, a, wide=1):
    """the rows (paths) of a per-path array that a slim record keeps (make_golden.one_iteration(slim=s): every s-th path,
    the dense float32 `dphi` every 4s-th); everything for a full record"""
    if 'slim_stride' not in z.files:
        return a
    return a[::int(z['slim_stride']) * wide]


def same_sample(z, key, got):
    import hashlib
    got = got.detach().numpy()
    assert np.array_equal(thin(z, got), z[key]), key
    if 'slim_stride' in z.files:    # the whole sample, bit for bit, through its SHA-1
        assert hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest() == str(z[key + '_sha1']), key


def first_sample(S):
    from src.dataset import Comb_loader
    s = S.setup
    domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
    pts = Comb_loader(s['N_r'], s['N_b'], domain, S.device)
    return domain, pts


@pytest.mark.parametrize('case', CASES)
def test_first_iteration_against_reference_vectors(golden_dir, case):
    _first_iteration(golden_dir, case)


@pytest.mark.parametrize('case', ['ref_plumb_midpoint', 'ref_d50_nt64_small_midpoint', 'ref_d100_small_midpoint', 'ref_wide_d6_midpoint',
                                  'ref_m1_d4_rk4'])
def test_first_iteration_with_the_hoisted_x_projection_against_reference_vectors(golden_dir, case, monkeypatch):
    """the same comparison with the test network's input layer split into xw_disc_xproj (spatial columns, once per path) and
    xw_disc_fwd_xproj -- on by itself from d = 45 (engine.xproj_min_d), forced here for the smaller ones"""
    from xnode_wan_pde_solver_amd.options import EngineOptions
    G = _first_iteration(golden_dir, case, options=EngineOptions(xproj_min_d=1))
    assert G.ptr('xproj') != 0 and float(G.xproj.abs().sum()) > 0


def _first_iteration(golden_dir, case, options=None):
    from utils.auxillary_funcs import L_norm, rel_err
    z, params = load(golden_dir, case)
    fname = str(params.pop('funcs', ''))
    general = fname[len('general_'):] if fname.startswith('general_') else ''
    F = P
    if general:       # the callables the fixture was recorded with (the reference's own classes ran them)
        import importlib.util
        import types
        spec = importlib.util.spec_from_file_location('general_funcs', os.path.join(golden_dir, 'general_funcs.py'))
        GF = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(GF)
        fa, fb, fc = GF.variant(general)
        F = types.SimpleNamespace(func_a=fa, func_b=fb, func_c=fc, func_h=P.func_h, func_f=P.func_f, func_g=P.func_g,
                                  func_u_sol=P.func_u_sol)
    S = make_solver(params, int(z['seed']), F=F, options=options)
    for tag, net in (('u', S.u_net), ('v', S.v_net)):
        sd = net.state_dict()
        assert list(sd.keys()) == [str(k) for k in z[tag + '_sd_keys']]
        for n, p in net.named_parameters():
            assert np.array_equal(p.detach().cpu().numpy(), z[tag + '_sd/' + n]), n
    domain, pts = first_sample(S)
    same_sample(z, 'x_u', pts.interioru[:, 0, 1:])
    same_sample(z, 'x_v', pts.interiorv[:, 0, 1:])
    same_sample(z, 'x_b', pts.boundary[:, 0, 1:])
    close(L_norm(pts.interioru, S.u_net, 2, P.func_u_sol, domain.V(), S.setup['N_r']), float(z['L2_start']), F32TOL)
    close(rel_err(pts.interioru, S.u_net, P.func_u_sol, 2, domain.V(), S.setup['N_r']), float(z['rel_start']), F32TOL)
    eng = S.engine
    if general == 'v1':       # every fused fast path is off: tabulated a_ij, b_i at t_0, c(u, t, x) through autograd, all inside the graphs
        assert not eng.structure.a_identity and not eng.structure.b_zero and eng.structure.c_kappa is None
    elif general == 'const':  # (round 5) one [d, d] matrix for all points, no b table, the linear reaction: the fused sub-step with A0
        assert not eng.structure.a_identity and eng.structure.b_zero and eng.structure.c_kappa == -0.7
    elif general == 'diag':   # (round 5) the diagonal [d, N] form of a, a b table, the linear reaction
        assert not eng.structure.a_identity and not eng.structure.b_zero and eng.structure.c_kappa == -0.7
    else:
        assert eng.structure.a_identity and eng.structure.b_zero and eng.structure.c_kappa == -1.0
    G = eng.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)   # host tensors: tabulated like the reference
    unames = [n for n, _ in S.u_net.named_parameters()]
    vnames = [n for n, _ in S.v_net.named_parameters()]

    def check(tag, which):
        close(thin(z, G.u.t()), z[tag + '/u'], F32TOL, F32TOL, tag + ' u')
        close(thin(z, G.v.t()), z[tag + '/v'], 1e-10 if tag == 'gen1' else F32TOL, 1e-12 if tag == 'gen1' else F32TOL, tag + ' v')
        close(thin(z, G.h), z[tag + '/h'], F32TOL, 1e-7)
        close(thin(z, G.f.t()), z[tag + '/f'], F32TOL, 1e-6)
        Gx = (G.gx + G.gs.unsqueeze(0) * G.ghT).t()
        close(thin(z, Gx), z[tag + '/Xgrad_l0'][:, 1:], 2e-5, 1e-6, tag + ' nabla_x u')
        dphi = z[tag + '/dphi']
        dphi0 = (G.w0.unsqueeze(0) * G.gxv + G.v[0].unsqueeze(0) * G.gwx0T).t()
        close(thin(z, dphi0, 4), dphi[:, 0, 1:], 2e-5, 1e-6, tag + ' nabla_x phi at t0')
        close(thin(z, (G.w.unsqueeze(0) * G.vt).t(), 4), dphi[:, :, 0], 2e-5, 1e-6, tag + ' d(phi)/dt')
        scal = eng.scal.cpu().numpy()
        if which == 'u':
            close(thin(z, G.g.t()), z[tag + '/g'], F32TOL, 1e-7)
            close(thin(z, G.ub.t()), z[tag + '/u_b'], F32TOL, F32TOL, tag + ' u_b')  # (boundary forward of this sub-step)
            close(scal[2] / G.N, float(z[tag + '/init']), 1e-5)
            close(scal[3] / (G.Nb * G.L), float(z[tag + '/bdry']), 1e-5)
            close(scal[4], float(z[tag + '/loss']), 1e-5, what=tag + ' loss_u')
            grad, names, blob = eng.grad_u, unames, eng.theta
        else:
            close(scal[5], float(z[tag + '/loss']), 1e-5, what=tag + ' loss_v')
            grad, names, blob = eng.grad_v, vnames, eng.phi
        gmax = max(float(np.abs(z[tag + '/grad/' + n]).max()) for n in names)
        for n, g_, p_ in zip(names, blob.split(grad), blob.params):
            close(g_, z[tag + '/grad/' + n], 1e-5, 1e-6 * gmax, tag + ' grad ' + n)
            close(p_, z[tag + '/after/' + n], 1e-5, 1e-7, tag + ' after ' + n)

    eng.generator_step(G)
    check('gen1', 'u')
    eng.generator_step(G)
    check('gen2', 'u')
    eng.discriminator_step(G)
    check('disc1', 'v')
    assert int(eng.adam_u['step'].item()) == 2 and int(eng.adam_v['step'].item()) == 1
    # value of I / int at the final parameters
    eng.use_graphs = False
    G.skip_v = False
    eng._disc_front(G)
    scal = eng.scal.cpu().numpy()
    close(scal[0], float(z['final/I']), 1e-5)
    close(np.log(scal[0] ** 2) - np.log(G.Vol * scal[1] / (G.N * G.L)), float(z['final/int']), 1e-5, 1e-6)
    # RNG stream position after the iteration's second sample
    from src.dataset import Comb_loader
    pts2 = Comb_loader(S.setup['N_r'], S.setup['N_b'], domain, S.device)
    same_sample(z, 'x_u_second', pts2.interioru[:, 0, 1:])
    close(L_norm(pts2.interioru, S.u_net, 2, P.func_u_sol, domain.V(), S.setup['N_r']), float(z['L2_end']), 1e-6)
    return G


def test_module_autograd_path_reproduces_reference_gradients(golden_dir):
    """user-facing path: u_net(X), v_net(XV), the `loss` class and .backward() -- gradients as Adam sees them"""
    from src.loss import loss
    from src.training import func_eval
    z, params = load(golden_dir, 'ref_tiny_midpoint')
    S = make_solver(params, int(z['seed']))
    domain, pts = first_sample(S)
    datau, datav, bdata = pts.interioru, pts.interiorv, pts.boundary   # host leaves, as on the reference's CPU path
    S.optimizer_u.zero_grad()
    pv, pu = S.v_net(datav), S.u_net(datau)
    close(pu.squeeze(2), z['gen1/u'], F32TOL, F32TOL)
    close(pv.squeeze(2), z['gen1/v'], 1e-10, 1e-12)
    h, f, g, a, b, c = func_eval(datau.clone().detach(), bdata.clone().detach(), S.setup, pu, P.func_a, P.func_b, P.func_c,
                                 P.func_h, P.func_f, P.func_g)
    Lo = loss(S.config['alpha'], a, b, c, h, f, g, S.setup, domain, S.device)
    lu = Lo.u(pu, pv, S.u_net, datau, datav, bdata)
    close(lu, float(z['gen1/loss']), 1e-5)
    lu.backward()
    names = [n for n, _ in S.u_net.named_parameters()]
    gmax = max(float(np.abs(z['gen1/grad/' + n]).max()) for n in names)
    for n, p in S.u_net.named_parameters():
        close(p.grad, z['gen1/grad/' + n], 1e-5, 1e-6 * gmax, 'autograd-path grad ' + n)
    S.optimizer_u.step()
    for n, p in S.u_net.named_parameters():
        close(p, z['gen1/after/' + n], 1e-5, 1e-7, 'autograd-path after ' + n)
    # single time slice at T0 returns [N, 1] like the reference (src/model.py:89-91)
    assert S.u_net(datau[:, :1, :].detach()).shape == (datau.shape[0], 1)
    # arbitrary point sets through v_net
    assert S.v_net(torch.rand(7, 3, S.setup['dim'] + 1)).shape == (7, 3, 1)


def test_module_autograd_path_with_adjoint_true():
    """u_net(X).backward() with config['adjoint'] = True: the module's autograd bridge asks the sweep for the continuous
    adjoint; x only receives the gradient that flows through the start value h(x)"""
    from oracle import refspec as R
    d = 3
    params = {'alpha': 1e3, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': True, 'solver': 'midpoint',
              'dim': d, 'N_t': 6, 'N_r': 21, 'N_b': 12, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
              'domain': 'Hypercube'}
    S = make_solver(params, 3)
    assert S.u_net.module.adjoint and S.engine.adjoint
    domain, pts = first_sample(S)
    X = pts.interioru.detach().clone().requires_grad_(True)
    g = torch.Generator().manual_seed(4)
    wgt = torch.randn(X.shape[0], X.shape[1], dtype=torch.float64, generator=g)
    (S.u_net(X).squeeze(2).cpu() * wgt).sum().backward()
    names = R.u_names(8)
    theta = {k_: dict(S.u_net.named_parameters())[n_].detach().cpu().clone().requires_grad_(True) for n_, k_ in names}
    Xo = pts.interioru.detach().clone().requires_grad_(True)
    (R.u_net(theta, params, Xo, P.func_h(Xo[:, 0, :])) * wgt).sum().backward()
    gmax = max(float(theta[k_].grad.abs().max()) for _, k_ in names)
    for n_, k_ in names:
        close(dict(S.u_net.named_parameters())[n_].grad, theta[k_].grad, 1e-6, 1e-8 * gmax, 'adjoint-mode grad ' + k_)
    close(X.grad[:, 0, 1:], Xo.grad[:, 0, 1:], 1e-5, 1e-7, 'nabla_x u (through the start value only)')


@pytest.mark.parametrize('m,q,adjoint', [(8, 9, False), (3, 4, False), (8, 9, True)])
def test_engine_against_oracle_general_coefficients(m, q, adjoint):
    """non-identity a, non-zero b, non-linear c(u): the structured fast paths are off, everything goes the general way.
    Second case: network depths other than the YAML's (the test network then always runs from its activation record).
    Third case: adjoint=True, the sweeps integrate torchdiffeq's continuous adjoint (oracle restatement, parity unpinned)"""
    from oracle import refspec as R
    d = 4
    params = {'alpha': 1e3, 'u_layers': m, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': q, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': adjoint, 'solver': 'midpoint',
              'dim': d, 'N_t': 9, 'N_r': 83, 'N_b': 45, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
              'domain': 'Hypercube'}

    def fa(X, i, j):
        return (1.0 + 0.5 * X[..., 1] ** 2) * (1.0 if i == j else 0.1 * torch.cos(X[..., 2]))

    def fb(X, i):
        return 0.3 * X[..., i + 1] * torch.exp(-X[..., 0])

    def fc(X, u):
        return -u ** 2 + 0.5 * X[..., 1:2] * 0 + torch.sin(X[..., 1:2])

    funcs = dict(FUNCS, a=fa, b=fb, c=fc)
    from src.training import NODE_WAN_solver
    torch.manual_seed(5)
    S = NODE_WAN_solver(params, fa, fb, fc, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
    st = S.engine.structure
    assert not st.a_identity and not st.b_zero and st.c_kappa is None
    torch.manual_seed(5)
    O = R.Solver(params, funcs, u_sol=P.func_u_sol, p=2)
    for n_, k_ in R.u_names(m):
        assert torch.equal(dict(S.u_net.named_parameters())[n_].detach().cpu(), O.theta[k_])
    rng = torch.get_rng_state()
    domain, pts = first_sample(S)
    torch.set_rng_state(rng)
    O.new_sample()
    assert torch.equal(O.X, pts.interioru.detach()) and torch.equal(O.BX, pts.boundary.detach())
    G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
    for step in ('u', 'u', 'v'):
        if step == 'u':
            o = O.generator_step()
            S.engine.generator_step(G)
            got, blob, names = S.engine.grad_u, S.engine.theta, R.u_names(m)
            close(S.engine.scal[4], o['loss'], 1e-8)
        else:
            o = O.discriminator_step()
            S.engine.discriminator_step(G)
            got, blob, names = S.engine.grad_v, S.engine.phi, R.V_NAME_MAP
            close(S.engine.scal[5], o['loss'], 1e-6)
        close(S.engine.scal[0], o['I'], 1e-5)
        gmax = max(float(o['grad'][k].abs().max()) for _, k in names)
        for (n_, k_), g_ in zip(names, blob.split(got)):
            close(g_, o['grad'][k_], 1e-5, 1e-6 * gmax, 'grad ' + k_)


@pytest.mark.parametrize('case,steps,windows,reached', [
    ('ref_traj_plumb_seed0_gpusem', 800, ((100, 200), (400, 600), (600, 800)), 0.01),
    ('ref_traj_d20_seed2_gpusem', 300, ((100, 200), (200, 300)), 0.02),
    ('ref_traj_d20_headline_seed4', 500, ((300, 400), (400, 500)), 0.015),
    # the other two fixed-grid schemes at the benchmarked size (30 outer iterations of the reference's own train() each)
    ('ref_traj_d20_headline_euler_seed5', 60, ((20, 40), (40, 60)), 0.4),
    ('ref_traj_d20_headline_rk4_seed6', 60, ((20, 40), (40, 60)), 0.1),
    # round 5: 25 outer iterations of the reference's own train() at widths of the generic path -- (48, 16) field, 100-wide test network
    ('ref_traj_generic_d3_seed14', 50, ((10, 30), (30, 50)), 0.06),
    # round 5: other sub-iteration counts than the YAML's (n1, n2) = (2, 1): (3, 2) over 20 outer iterations, (1, 3) over 30
    ('ref_traj_n1_3_n2_2_d3_seed16', 60, ((20, 40), (40, 60)), 0.08),
    ('ref_traj_n1_1_n2_3_d3_seed17', 30, ((10, 20), (20, 30)), 0.45),
    # round 5: [0.25, 1.5] x [-0.5, 1.5]^3, 15 outer iterations
    ('ref_traj_interval_d3_seed21', 30, ((10, 20), (20, 30)), 0.2),
    # round 5: general a_ij(t, x), b_i(t, x), c(u, t, x) through the reference's own train(), 20 outer iterations at alpha = 1e3
    ('ref_traj_general_d3_seed36', 40, ((10, 25), (25, 40)), 1.0),
    # round 5: a hook that draws random numbers itself (12 outer iterations): the loop's own draws stay where the reference makes them
    ('ref_traj_hook_draws_d3_seed39', 24, ((6, 15), (15, 24)), 1.0)])
def test_trained_error_trajectory_matches_reference(golden_dir, tmp_path, case, steps, windows, reached):
    """BASELINE config 1 (d=5, N_r=256, N_b=64, N_t=16; seed 0, 400 outer iterations = 800 generator sub-steps), the
    headline dimension (d=20, N_r=128, N_b=96, N_t=12; seed 2, 150 outer iterations) and -- round 4 -- BASELINE configs[1] AT THE
    BENCHMARKED SIZE (d=20, N_r=N_b=4096, N_t=32; seed 4, 250 outer iterations of the reference's own train(), 25 min of its CPU time:
    rel-L2 0.78 -> 0.0069, under its own acceptance rule of 0.01)
    through train(): rel-L2 logged by the `stop` hook at every sub-step, compared with the reference's own runs (fixtures)."""
    from utils.auxillary_funcs import rel_err
    z, params = load(golden_dir, case)
    ref = z['rel_l2']
    log = []
    F = P
    if params.pop('funcs', None) == 'general_v1':      # (round 5: general a_ij, b_i, c through train(): tests/golden/general_funcs.py)
        F = _general_funcs(golden_dir, P, with_b=True)

    def hook(self, pts, domain):
        log.append(float(rel_err(pts, self.u_net, self.func_u_sol, self.p, domain.V(), self.params['N_r'])))
        if 'hook_draws' in case:                   # (the fixture's hook consumed both global generators at every call)
            torch.rand(3)
            np.random.rand(2)
        return False
    S = make_solver(params, int(z['seed']), F=F, stop=hook)
    S.tabulate_on_host = True          # tabulate h, f, g like the reference's CPU run (tight early-step comparison)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        S.train(report=False)
    finally:
        os.chdir(cwd)
    got = np.array(log)
    out = os.environ.get('XW_DUMP_TRAJ')
    if out:
        np.savez(out + '_' + case, got=got, ref=ref)
    assert got.shape == ref.shape == (steps,)
    # same seeds, same arithmetic: the runs track each other closely before chaotic decorrelation sets in
    np.testing.assert_allclose(got[:50], ref[:50], rtol=2e-3)
    if case in ('ref_traj_d20_headline_euler_seed5', 'ref_traj_d20_headline_rk4_seed6', 'ref_traj_generic_d3_seed14',
                'ref_traj_n1_3_n2_2_d3_seed16', 'ref_traj_n1_1_n2_3_d3_seed17', 'ref_traj_interval_d3_seed21',
                'ref_traj_general_d3_seed36', 'ref_traj_hook_draws_d3_seed39'):
        np.testing.assert_allclose(got, ref, rtol=1e-4)               # (60 logged values: before the decorrelation sets in)
    if case == 'ref_traj_d20_headline_seed4':
        # 4096 paths per sample: the two runs stay together for the first 60 outer iterations (measured: 2e-6 at worst over the
        # first 120 logged values, 1e-5 at 140, 8e-4 at 160, then the usual exponential decorrelation of two adversarial runs: a
        # loss spike falls on step 287 here and on step ~360 there); after 250 outer iterations 0.0101 against the reference's 0.0069
        np.testing.assert_allclose(got[:120], ref[:120], rtol=1e-4)
    # north-star criterion: trained relative-L2 error within 1e-2 absolute of the reference's, on windowed statistics
    for lo, hi in windows:
        assert abs(np.median(got[lo:hi]) - np.median(ref[lo:hi])) < 1e-2, (lo, hi, np.median(got[lo:hi]), np.median(ref[lo:hi]))
    # (the long run at the benchmarked size: its last VALUE can sit on one of the loss spikes of adversarial training -- the
    #  reference's own run has one at step ~360 --, so the end of the run is compared on the median of its last 20 values)
    tail = 20 if case == 'ref_traj_d20_headline_seed4' else 1
    assert abs(np.median(got[-tail:]) - np.median(ref[-tail:])) < 1e-2
    assert got[windows[-1][0]:].min() < reached and ref[windows[-1][0]:].min() < reached
    d = params['dim']
    for fn in ('losses_NODE_%d.json' % d, 'L2_NODE_%d.json' % d, 'Time_NODE_%d.json' % d, 'best_model_weights_NODE.pth'):
        assert (tmp_path / fn).exists(), fn
    sd = torch.load(tmp_path / 'best_model_weights_NODE.pth')
    assert ('module.ODE_rhs.net.%d.weight' % (2 * (params['u_layers'] - 1))) in sd
    assert sd['module.final_linear.weight'].shape == (1, params['u_hidden_dim'])


@pytest.mark.parametrize('case,name', [('ref_cone_groups', 'NSphere_TCone'), ('ref_hourglass_groups', 'NSphere_THourglass'),
                                       ('ref_cone_ex43_d10_groups', 'NSphere_TCone'),
                                       ('ref_hourglass_ex43_d10_groups', 'NSphere_THourglass'),
                                       ('ref_cone_r07_groups', 'NSphere_TCone'),      # (round 5: radius 0.7, a one-path boundary group)
                                       # round 5: general a_ij(t, x), c(u, t, x) on the list domains
                                       ('ref_cone_general_groups', 'NSphere_TCone'), ('ref_hourglass_general_groups', 'NSphere_THourglass'),
                                       # round 5: alpha = 1
                                       ('ref_cone_alpha1_groups', 'NSphere_TCone'), ('ref_hourglass_alpha1_groups', 'NSphere_THourglass'),
                                       ('ref_hourglass_alpha1_general_groups', 'NSphere_THourglass'),
                                       # round 5: one constant matrix a, c = -0.7 u -- the group runner's fused path with an A0 table
                                       ('ref_hourglass_const_a_groups', 'NSphere_THourglass')])
def test_sphere_domain_groups_against_reference_vectors(golden_dir, case, name):
    """time-varying ball domains (BASELINE config 5 family): float64 groups of different lengths, late-entry groups that
    start on the moving boundary (g start values), time-dependent weight w, single-time boundary groups, and the
    gradient carried across the groups of one sub-iteration -- against vectors recorded from the reference"""
    from src.dataset import Comb_loader
    from utils.auxillary_funcs import L_norm
    z, params = load(golden_dir, case)
    # the last two cases are BASELINE configs[4]: the Ex4_3 problem (configs/Ex4_3_funcs.py:6-49 of the reference: every
    # coordinate enters u_sol, c = -u) at d = 10, recorded from the reference with its own Ex4_3 callables
    F = P
    fname = params.pop('funcs', 'Ex4_1_funcs')
    if fname.startswith('Ex4_3_funcs'):
        import configs.Ex4_3_funcs as F
    if fname.endswith('+general_ac'):        # (round 5: general a_ij(t, x) and c(u, t, x), b = 0 -- tests/golden/general_funcs.py)
        import importlib.util
        import types
        spec = importlib.util.spec_from_file_location('general_funcs', os.path.join(golden_dir, 'general_funcs.py'))
        GF = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(GF)
        F = types.SimpleNamespace(func_a=GF.func_a, func_b=F.func_b, func_c=GF.func_c, func_h=F.func_h, func_f=F.func_f, func_g=F.func_g,
                                  func_u_sol=F.func_u_sol)
    if fname.endswith('+general_const'):     # (one constant matrix a, c = -0.7 u: the fused path / the group runner with an A0 table)
        import importlib.util
        import types
        spec = importlib.util.spec_from_file_location('general_funcs', os.path.join(golden_dir, 'general_funcs.py'))
        GF = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(GF)
        F = types.SimpleNamespace(func_a=GF.const_a, func_b=F.func_b, func_c=GF.lin_c, func_h=F.func_h, func_f=F.func_f, func_g=F.func_g,
                                  func_u_sol=F.func_u_sol)
    assert params['domain'] == name
    S = make_solver(params, int(z['seed']), F=F)
    s = S.setup
    domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
    pts = Comb_loader(s['N_r'], s['N_b'], domain, S.device)
    assert len(pts.interioru) == int(z['n_interior']) and len(pts.boundary) == int(z['n_boundary'])
    # (bit-exact in this container, tests/test_host_logic.py; the GPU box's host CPU rounds a few of the float64 entry
    #  times of late-entry paths differently in the last bit)
    for k, g_ in enumerate(pts.interioru):
        close(g_, z['interior/%d' % k], 1e-13, 1e-15, 'interior group %d' % k)
    for k, g_ in enumerate(pts.boundary):
        close(g_, z['boundary/%d' % k], 1e-13, 1e-15, 'boundary group %d' % k)
    # the reference's diagnostic on a list domain: `u_net(x).squeeze()` against func_u_sol(x) [N,1] on the single-slice group
    # is the [N,N] table over all pairs (utils/auxillary_funcs.py:19) -- reproduced
    from utils.auxillary_funcs import rel_err
    close(L_norm(pts.interioru, S.u_net, 2, F.func_u_sol, domain.V(), s['N_r']), float(z['L2_start']), 1e-7, what='L_norm on the groups')
    close(rel_err(pts.interioru, S.u_net, F.func_u_sol, 2, domain.V(), s['N_r']), float(z['rel_start']), 1e-7, what='rel_err on the groups')
    eng = S.engine
    # the triples the reference's loop visits: (k, k) for every k the loader yields, INCLUDING the single-slice group 0 at
    # T0 (59 % of the interior paths at d = 10) with its [N,N] pairwise loss terms and the T0 boundary group
    pairs = [tuple(p) for p in z['pairs']]
    assert pairs == [(k, k) for k in range(min(len(pts.interioru), len(pts.boundary)))] == [(k, k) for k in range(len(S._groups(pts)))]
    groups = [eng.load_group(pts.interioru[ki], pts.interiorv[ki], pts.boundary[kb], domain) for ki, kb in pairs]
    assert groups[0].L == 1 and groups[0].pair_i and groups[0].pair_b and not any(G.pair_i for G in groups[1:])
    for G in groups:
        G.persistent = False
    step = 0
    for which in ('u', 'u', 'v'):
        eng.begin_substep(which, True)
        for G in groups:
            tag = 'step%d' % step
            assert str(z[tag + '/which']) == which
            if which == 'u':
                eng.generator_step(G)
                got_loss, got_grad, blob = eng.scal[4], eng.grad_u, eng.theta
            else:
                eng.discriminator_step(G)
                got_loss, got_grad, blob = eng.scal[5], eng.grad_v, eng.phi
            close(G.u.t(), z[tag + '/u'], 1e-5, 1e-7, tag + ' u')
            close(G.v.t(), z[tag + '/v'], 1e-5, 1e-7, tag + ' v')
            close(got_loss, float(z[tag + '/loss']), 1e-5, what=tag + ' loss')
            if which == 'u':
                init = float(eng.scal[2]) / G.Nglob + G.init_off
                bdry = float(eng.scal[3]) / (G.Nbglob * G.Lb) + G.bdry_off
                close(init, float(z[tag + '/init']), 1e-9, what=tag + ' init penalty')
                close(bdry, float(z[tag + '/bdry']), 1e-9, what=tag + ' boundary penalty')
            ref = z[tag + '/grad']
            close(got_grad, ref, 1e-5, 1e-6 * float(np.abs(ref).max()), tag + ' grad (carried over the groups)')
            close(blob.data, z[tag + '/after'], 1e-5, 1e-7, tag + ' params after Adam')
            step += 1
    assert step == int(z['n_steps'])


@pytest.mark.parametrize('case,steps', [('ref_traj_cone_ex43_d3_seed0', 200), ('ref_traj_hourglass_ex43_d3_seed1', 120),
                                        ('ref_traj_cone_ex43_d10_full_seed2', 16), ('ref_traj_hourglass_ex43_d10_full_seed3', 16),
                                        ('ref_traj_hourglass_ex43_d3_euler_seed7', 80), ('ref_traj_cone_ex43_d3_rk4_seed8', 80),
                                        # round 5: n1 = 3, n2 = 2 (12 outer iterations): several discriminator sub-iterations move phi
                                        # between them, three generator sub-iterations see the same phi
                                        ('ref_traj_cone_n1_3_n2_2_d3_seed18', 36),
                                        # round 5: radius 0.7, 10 outer iterations each
                                        ('ref_traj_cone_r07_d3_seed22', 20), ('ref_traj_hourglass_r07_d3_seed23', 20),
                                        # round 5: general a_ij, c(u, t, x), 8 outer iterations each at alpha = 1e2
                                        ('ref_traj_cone_general_d3_seed37', 16), ('ref_traj_hourglass_general_d3_seed38', 16),
                                        # round 5: a hook that draws random numbers itself
                                        ('ref_traj_cone_hook_draws_d3_seed40', 16)])
def test_ball_domain_training_trajectory_follows_reference(golden_dir, tmp_path, case, steps):
    """BASELINE configs[4] family, through train(): NSphere_TCone (seed 0, 100 outer iterations) and NSphere_THourglass (seed 1,
    60), Ex4_3, d = 3, N_r = 256, N_b = 128, N_t = 10 -- and, round 4, both domains AT THE CONFIG'S STATED SIZE (d = 10,
    N_r = N_b = 8192, N_t = 20, alpha = 1e4; 8 outer iterations of the reference's own train(), 2 min of its CPU time each; and the other two fixed-grid schemes on list domains: euler on the hourglass, rk4 on the cone, 40 outer iterations each:
    11-12 and 18-20 groups per sample, single-slice groups of ~3600 paths with their [N, N] pairwise terms): the natural group loop (single-slice T0 groups with the reference's pairwise terms, Adam
    skipping the field's parameters there).  The `stop` hook evaluates u_theta on the fixture's fixed multi-slice probe
    group (the reference's own L_norm is all-pairs on list domains).  The REFERENCE DOES NOT CONVERGE on the ball domains
    (as run on this software stack its probe error grows from 1.4 to > 100, for alpha = 1e2 .. 1e8, both domains, Ex4_1
    and Ex4_3: DESIGN 8), what is pinned is that the engine FOLLOWS the reference's run, sub-iteration by sub-iteration."""
    import configs.Ex4_3_funcs as F
    z, params = load(golden_dir, case)
    if params.pop('funcs').endswith('+general_ac'):
        F = _general_funcs(golden_dir, F, with_b=False)
    ref = z['rel_l2']
    probe, sol = torch.from_numpy(z['probe']), torch.from_numpy(z['probe_sol'])
    log = []

    def hook(self, pts, domain):
        with torch.no_grad():
            up = self.u_net(probe).squeeze(2).cpu()
        log.append(float(torch.sqrt(torch.mean((up - sol) ** 2) / torch.mean(sol ** 2))))
        if 'hook_draws' in case:                   # (the fixture's hook consumed both global generators at every call)
            torch.rand(3)
            np.random.rand(2)
        return False
    S = make_solver(params, int(z['seed']), F=F, stop=hook)
    S.tabulate_on_host = True
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        S.train(report=False)
    finally:
        os.chdir(cwd)
    got = np.array(log)
    out = os.environ.get('XW_DUMP_TRAJ')
    if out:
        np.savez(out, got=got, ref=ref)
    assert got.shape == ref.shape == (steps,)
    # Ball-domain samples are float64 end to end (src/dataset.py:65-96), so nothing rounds to float32 on the way and the two
    # runs do not decorrelate over these 100 outer iterations (~2400 optimiser steps over all groups): measured on the
    # MI355X 1.8e-9 relative at worst.  North-star criterion (trained error within 1e-2 absolute of the reference's on the
    # same seeds): holds at EVERY logged sub-iteration, not only at the end.
    # (radius 0.7, seed 22: this run is ill-conditioned from its fourth outer iteration on -- the engine against ITSELF with the narrow
    #  tiles off, a change of summation order only, is 3e-15 / 2e-11 / 9e-6 apart at logged values 0 / 4 / 6 and 4e-5 later, exactly
    #  the engine's distance from the reference there: tools/traj_r07_sensitivity.py -- so that fixture is followed to 2e-4)
    np.testing.assert_allclose(got, ref, rtol=2e-4 if case == 'ref_traj_cone_r07_d3_seed22' else 1e-6)
    np.testing.assert_allclose(got[:4], ref[:4], rtol=1e-7)
    assert np.abs(got - ref).max() < 1e-2
    late = min(20, steps // 2)
    assert got[late:].min() > 1.0 and ref[late:].min() > 1.0      # (the run the reference produces here does not converge)


def test_pipelined_loop_flushes_its_last_iteration_when_interrupted(golden_dir, tmp_path):
    """the pipelined train() writes iteration k's files while the GPU works on k + 1: an exception inside the loop must still
    leave every COMPUTED iteration on disk (losses, L2, best weights) -- the reference writes before it moves on"""
    z, params = load(golden_dir, 'ref_plumb_midpoint')
    S = make_solver(dict(params, iterations=6), int(z['seed']))
    calls = {'n': 0}
    real = S.engine.discriminator_step

    def flaky(G):
        calls['n'] += 1
        if calls['n'] == 4:
            raise KeyboardInterrupt()
        return real(G)
    S.engine.discriminator_step = flaky
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        with pytest.raises(KeyboardInterrupt):
            S.train(report=False)
        losses = json.load(open('losses_NODE_%d.json' % params['dim']))
        assert len(losses) == 3 * params['n1'] and all(np.isfinite(losses))       # iterations 0, 1, 2 were complete
        assert os.path.exists('best_model_weights_NODE.pth') and os.path.exists('L2_NODE_%d.json' % params['dim'])
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize('name,general,hoist,widths', [('NSphere_TCone', False, False, None), ('NSphere_THourglass', False, False, None),
                                                       ('NSphere_TCone', True, False, None), ('NSphere_THourglass', True, False, None),
                                                       ('NSphere_THourglass', False, True, None),
                                                       ('NSphere_THourglass', False, False, (48, 16, 100)),     # (the wide containers)
                                                       ('NSphere_TCone', True, False, (64, 16, 70))])
def test_group_substep_runner_matches_the_launch_by_launch_path(tmp_path, name, general, hoist, widths, monkeypatch):
    """xw_substep_gen / xw_substep_disc (one C-ABI call per group sub-step, csrc/xw_substep.hip) against the same chain issued
    launch by launch from engine.py: bit-identical parameters after three outer iterations over all groups of a ball domain
    (single-slice pairwise groups, boundary groups on their own grids, point-mode test network, carried gradients).
    general: non-identity a_ij(t, x) and the linear reaction c = -0.7 u (the A0 table and xw_weak_contract_general inside the call;
    XW_ELEMENTWISE_SINGLE_SLICE semantics are not involved: b = 0 keeps the pairwise groups allowed)
    hoist: every path-mode group with the x-projection table in front of its test network (XwGroup.xproj), in both forms
    widths: other network widths -- the wide containers of round 6 (the stepper's duo sweep on 16x16x4 tiles, the 96- / 128-wide
    test network) under the group runner"""
    from xnode_wan_pde_solver_amd.options import EngineOptions
    opts = EngineOptions(xproj_min_d=1) if hoist else None
    F = P
    if general:
        class F:      # noqa: N801
            func_h, func_f, func_g, func_u_sol, func_b = P.func_h, P.func_f, P.func_g, P.func_u_sol, P.func_b

            @staticmethod
            def func_a(X, i, j):
                return (1.0 + 0.5 * X[..., 1] ** 2) * (1.0 if i == j else 0.1 * torch.cos(X[..., 2]))

            @staticmethod
            def func_c(X, u):
                return -0.7 * u
    params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 4, 'N_t': 8, 'N_r': 300, 'N_b': 200, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 3, 'domain': name}
    if widths is not None:
        params.update(u_hidden_dim=widths[0], u_hidden_hidden_dim=widths[1], v_hidden_dim=widths[2])
    out = []
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for runner in (True, False):
            S = make_solver(params, 3, F=F, options=opts)
            assert S.engine.structure.a_identity != general and S.engine.structure.c_kappa == (-0.7 if general else -1.0)
            S.engine.use_runner = runner
            losses = S.train(report=False)
            out.append((S.engine.theta.data.clone(), S.engine.phi.data.clone(), list(losses)))
    finally:
        os.chdir(cwd)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    assert torch.isfinite(out[0][0]).all()


def test_sphere_domain_trains_end_to_end(tmp_path):
    """train() over a list domain: group protocol, truncation, single-slice groups, per-group optimiser steps"""
    params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 4, 'N_t': 8, 'N_r': 300, 'N_b': 200, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 3,
              'domain': 'NSphere_THourglass'}
    S = make_solver(params, 3)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        losses = S.train(report=False)
    finally:
        os.chdir(cwd)
    assert len(losses) == 6 and all(np.isfinite(losses))
    assert torch.isfinite(S.engine.theta.data).all() and torch.isfinite(S.engine.phi.data).all()
    assert int(S.engine.adam_u['step'].item()) > 6            # one optimiser step per group, not per sub-iteration


def test_test_net_reuse_is_exact(golden_dir):
    """opt-in reuse of v, dv/dt, nabla_x v(t_0) while phi and the sample are unchanged gives bit-identical parameters"""
    z, params = load(golden_dir, 'ref_plumb_midpoint')
    outs = []
    for reuse in (False, True):
        S = make_solver(params, 0)
        domain, pts = first_sample(S)
        S.engine.reuse_test_net = reuse
        G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        for k in range(2):
            S.engine.generator_step(G)
            S.engine.generator_step(G)
            S.engine.discriminator_step(G)
            G = S.engine.load_group(pts.interiorv, pts.interioru, pts.boundary, domain, into=G)      # "resample" in place
        outs.append((S.engine.theta.data.clone(), S.engine.phi.data.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_reused_test_net_is_dropped_when_phi_moves_outside_the_engine(golden_dir):
    """cached v must not survive in-place edits of the parameters (load_state_dict, optimizer_v.step(), ...)"""
    z, params = load(golden_dir, 'ref_tiny_midpoint')
    thetas = []
    for reuse in (True, False):
        S = make_solver(params, 7)
        domain, pts = first_sample(S)
        S.engine.reuse_test_net = reuse
        G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        S.engine.generator_step(G)                               # leaves v of the current phi in the group's buffers
        with torch.no_grad():
            for p_ in S.v_net.parameters():
                p_.mul_(1.25)                                    # torch-side write: bumps the parameters' version counters
        S.engine.generator_step(G)
        thetas.append(S.engine.theta.data.clone())
    assert torch.equal(thetas[0], thetas[1])


def test_activation_stores_do_not_change_the_step(golden_dir):
    """sweeps / test-network backward reading the stored layer inputs (default) == recomputing them (keep_activations off)"""
    z, params = load(golden_dir, 'ref_plumb_midpoint')
    outs = []
    for keep in (True, False):
        S = make_solver(params, 0)
        domain, pts = first_sample(S)
        S.engine.keep_activations = keep
        G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        assert (G.act is not None) == keep and (G.vact is not None) == keep
        S.engine.generator_step(G)
        S.engine.generator_step(G)
        S.engine.discriminator_step(G)
        outs.append((S.engine.theta.data.clone(), S.engine.phi.data.clone(), S.engine.scal.clone()))
    close(outs[0][0], outs[1][0], 1e-9, 1e-12, 'theta')
    close(outs[0][1], outs[1][1], 1e-9, 1e-12, 'phi')
    close(outs[0][2][:6], outs[1][2][:6], 1e-10, 0.0, 'sums and losses')


@pytest.mark.parametrize('case', ['ref_plumb_midpoint', 'ref_d20_small_midpoint'])
def test_compact_and_wide_generator_schedules_give_the_same_bits(golden_dir, case):
    """Engine._gen_front_compact (groups of at most `compact_tiles` tiles: all three sweep jobs in ONE launch on the main stream, the
    slabs summed by the update) against the wide schedule (sweeps A + boundary on a side stream, sweep B behind the test network, the
    slabs of the former summed early): the same kernels on the same arguments, the same summation trees -- bit-identical parameters,
    gradients and sums after g, g, d, g"""
    z, params = load(golden_dir, case)
    outs = []
    for tiles in (10 ** 6, 0):
        S = make_solver(params, 0)
        domain, pts = first_sample(S)
        eng = S.engine
        eng.compact_tiles = tiles
        G = eng.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        assert eng._compact(G, True, True) == (tiles > 0)
        for kind in 'ggdg':
            (eng.generator_step if kind == 'g' else eng.discriminator_step)(G)
        outs.append((eng.theta.data.clone(), eng.phi.data.clone(), eng.scal.clone(), eng.grad_u.clone(), sorted(G.graphs)))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[4] == b[4]


def test_evaluation_off_the_boundary_matches_reference(golden_dir):
    """u_net on paths that start neither at T0 nor on the boundary: bound_pad / fillt densified grid (src/model.py:92-106)"""
    z, params = load(golden_dir, 'ref_boundpad')
    S = make_solver(params, int(z['seed']))
    for k in range(int(z['n'])):
        X = torch.from_numpy(z['%d/X' % k])
        with torch.no_grad():
            u = S.u_net(X)
        assert tuple(u.shape) == z['%d/u' % k].shape
        close(u, z['%d/u' % k], F32TOL, F32TOL, 'case %d' % k)


def test_proj_saves_what_the_reference_saves(golden_dir, tmp_path):
    """utils.auxillary_funcs.proj (reference :34-98, the evaluation path's plotting helper): guess_cn.npy / error_cn.npy and the
    picture for a (t, x_1) slice, an (x_1, x_2) slice at the fixed time T (every row starts off T0: bound_pad / fillt inside
    u_net) and a (t, x_3) slice -- against the arrays the reference's own proj() saved for the same initial weights"""
    from utils.auxillary_funcs import proj
    z, params = load(golden_dir, 'ref_proj')
    S = make_solver(params, int(z['seed']))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for k in range(int(z['n'])):
            axes = [int(a) for a in z['%d/axes' % k]]
            proj(S.u_net, S.setup, 7, S.device, axes=axes, resolution=12, colours=6, save=True, show=False, func_u_sol=P.func_u_sol)
            close(np.load('guess_cn.npy'), z['%d/guess' % k], F32TOL, F32TOL, 'guess along %s' % axes)
            close(np.load('error_cn.npy'), z['%d/error' % k], F32TOL, 10 * F32TOL, 'error along %s' % axes)
            assert os.path.getsize('plot_at_7_along_' + str(axes) + '.png') > 1000
    finally:
        os.chdir(cwd)
    # L_norm / rel_err (reference utils/auxillary_funcs.py:7-30) at p = 1, 2, 3: one tensor (cube) and a list of groups (cone)
    from utils.auxillary_funcs import L_norm, rel_err
    from src.dataset import Comb_loader, Hypercube, NSphere_TCone
    s_ = S.setup
    for tag, shape in (('cube', Hypercube(params['shape_param'], s_['dim'], 0, 1, s_['N_t'])), ('cone', NSphere_TCone(1.0, s_['dim'], 0, 1, s_['N_t']))):
        torch.manual_seed(21)
        np.random.seed(21)
        pts = Comb_loader(40, 24, shape, S.device)
        for p_ in (1, 2, 3):
            with torch.no_grad():
                close(L_norm(pts.interioru, S.u_net, p_, P.func_u_sol, shape.V(), 40), float(z['%s/L%d' % (tag, p_)]), 2e-6, what='%s L%d' % (tag, p_))
                close(rel_err(pts.interioru, S.u_net, P.func_u_sol, p_, shape.V(), 40), float(z['%s/rel%d' % (tag, p_)]), 2e-6, what='%s rel%d' % (tag, p_))


def test_evaluation_off_the_boundary_on_the_hourglass_matches_reference(golden_dir):
    """the per-path padded grids of NSphere_THourglass.bound_pad (src/dataset.py:127-152): buckets by grid length, every
    bucket on the grid of its first path, results ordered bucket by bucket -- on the inputs the reference survives"""
    z, params = load(golden_dir, 'ref_boundpad_hourglass')
    S = make_solver(params, int(z['seed']))
    for k in range(int(z['n'])):
        X = torch.from_numpy(z['%d/X' % k])
        with torch.no_grad():
            u = S.u_net(X)
        assert tuple(u.shape) == z['%d/u' % k].shape
        close(u, z['%d/u' % k], F32TOL, F32TOL, 'case %d' % k)


def test_evaluation_off_the_boundary_on_the_cone_matches_reference(golden_dir):
    """NSphere_TCone.bound_pad (src/dataset.py:220-223: one padded grid for the whole batch, from its first path's times) -- on the
    inputs the reference survives"""
    z, params = load(golden_dir, 'ref_boundpad_cone')
    S = make_solver(params, int(z['seed']))
    assert int(z['n']) >= 4
    for k in range(int(z['n'])):
        X = torch.from_numpy(z['%d/X' % k])
        with torch.no_grad():
            u = S.u_net(X)
        assert tuple(u.shape) == z['%d/u' % k].shape
        close(u, z['%d/u' % k], F32TOL, F32TOL, 'case %d' % k)


def test_checkpoint_resume_is_bit_exact(golden_dir, tmp_path):
    """train 2+2 outer iterations with a save/load in the middle == train 4 outer iterations in one go"""
    z, params = load(golden_dir, 'ref_tiny_midpoint')
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        A = make_solver(dict(params, iterations=4), 5)
        A.train()
        B = make_solver(dict(params, iterations=2), 5)
        B.train()
        B.save_checkpoint('mid.pt')
        C = make_solver(dict(params, iterations=2), 99)          # different init: everything must come from the file
        C.load_checkpoint('mid.pt')
        C.train()
    finally:
        os.chdir(cwd)
    assert torch.equal(A.engine.theta.data, C.engine.theta.data) and torch.equal(A.engine.phi.data, C.engine.phi.data)
    assert int(C.engine.adam_u['step'].item()) == 8 and int(C.engine.adam_v['step'].item()) == 4


def test_notebook_flow_runs_end_to_end(tmp_path, monkeypatch):
    """examples/notebook_flow.py = the reference's example.ipynb as a script (imports of cell 0, params dict of cell 10
    without shape_param, train(report=True) of cell 11 with the contour plots written to files): runs unchanged and
    the error on a fresh sample has dropped well below the untrained network's"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('notebook_flow', os.path.join(os.path.dirname(__file__), '..', 'examples',
                                                                               'notebook_flow.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    solver, err = mod.main(iterations=120, report_it=60, show_plt=False, workdir=str(tmp_path))
    assert err < 0.3
    files = set(os.listdir(tmp_path))
    assert {'losses_NODE_5.json', 'L2_NODE_5.json', 'Time_NODE_5.json', 'best_model_weights_NODE.pth'} <= files
    assert any(f.endswith('.png') for f in files), files


def test_sampling_drawn_ahead_changes_nothing(tmp_path, monkeypatch):
    """train() draws the next host samples on a helper thread while the GPU works (solver.overlap_sampling): the losses,
    the parameters and the position of both random generators afterwards are those of the in-line loop"""
    monkeypatch.chdir(tmp_path)
    params = {'alpha': 1e3, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 4, 'N_t': 8, 'N_r': 96, 'N_b': 48, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 7,
              'domain': 'Hypercube'}
    out = []
    for ahead in (True, False):
        S = make_solver(params, 11)
        S.overlap_sampling = ahead
        losses = list(S.train())
        out.append((losses, S.engine.theta.data.clone(), S.engine.phi.data.clone(), torch.get_rng_state().clone(),
                    np.random.get_state()[1].copy(), open('losses_NODE_4.json').read(), open('L2_NODE_4.json').read()))
    a, b = out
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.equal(a[3], b[3]) and (a[4] == b[4]).all() and a[5] == b[5] and a[6] == b[6]


def test_general_contraction_kernel_forms():
    """xw_weak_contract_general against torch.einsum for the four forms of a (identity, one matrix, diagonal, full table)
    with and without b, at a size with a ragged last block"""
    from xnode_wan_pde_solver_amd import kernels as KN
    d, N = 7, 1000
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).cuda()   # noqa: E731
    gx, ghT, gxv, gwx0T, gs, w0, v0 = r(d, N), r(d, N), r(d, N), r(d, N), r(N), r(N), r(N)
    Gx, dphi = gx + gs * ghT, w0 * gxv + v0 * gwx0T
    full, diag, const, B0 = r(d, d, N), r(d, N), r(d, d), r(d, N)
    for amode, A0, ref in ((0, None, (dphi * Gx).sum(0)), (1, const, torch.einsum('ij,in,jn->n', const, dphi, Gx)),
                           (2, diag, (diag * dphi * Gx).sum(0)), (3, full, torch.einsum('ijn,in,jn->n', full, dphi, Gx))):
        for B in (None, B0):
            out = torch.empty(N, dtype=torch.float64, device='cuda')
            KN.weak_contract_general(A0, amode, B, gx, gs, ghT, gxv, w0, gwx0T, v0, out)
            want = ref if B is None else ref + v0 * w0 * (B * Gx).sum(0)
            close(out, want, 1e-12, 1e-12)
    # N == d: a constant [d,d] matrix and a diagonal [d,N] table have the same shape -- the form is stated, not guessed
    sq, o1, o2 = r(d, d), torch.empty(d, dtype=torch.float64, device='cuda'), torch.empty(d, dtype=torch.float64, device='cuda')
    a_ = [t_[:, :d].contiguous() for t_ in (gx, ghT, gxv, gwx0T)]
    b_ = [t_[:d].contiguous() for t_ in (gs, w0, v0)]
    KN.weak_contract_general(sq, 1, None, a_[0], b_[0], a_[1], a_[2], b_[1], a_[3], b_[2], o1)
    KN.weak_contract_general(sq, 2, None, a_[0], b_[0], a_[1], a_[2], b_[1], a_[3], b_[2], o2)
    Gs, ds = a_[0] + b_[0] * a_[1], b_[1] * a_[2] + b_[2] * a_[3]
    close(o1, torch.einsum('ij,in,jn->n', sq, ds, Gs), 1e-12, 1e-12)
    close(o2, (sq * ds * Gs).sum(0), 1e-12, 1e-12)


def test_general_coefficients_are_captured_and_batched():
    """general a written with tensor operations: ONE batched tabulation call, exact compression (diagonal / constant
    forms), the contraction kernel inside the captured graph, and a black-box c(u) replayed from the graph too"""
    from src.training import NODE_WAN_solver
    d = 5
    params = {'alpha': 1e3, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': d, 'N_t': 7, 'N_r': 130, 'N_b': 60, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
              'domain': 'Hypercube'}
    calls = {'a': 0}

    def fa(X, i, j):                          # tensor-friendly: works with index tensors (diagonal, varies with x)
        calls['a'] += 1
        return (1.0 + 0.5 * X[..., 1] ** 2) * (i == j)

    def fb(X, i):
        return 0.3 * X[..., 0] * (i + 1.0)

    def fc(X, u):
        return -u ** 2 + torch.sin(X[..., 1:2])

    outs = []
    for graphs in (True, False):
        torch.manual_seed(5)
        S = NODE_WAN_solver(params, fa, fb, fc, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
        S.engine.use_graphs = graphs
        domain, pts = first_sample(S)
        calls['a'] = 0
        G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        assert calls['a'] <= 1 + 3                                  # one batched call + the three spot checks
        assert tuple(G.A0.shape) == (d, G.N) and tuple(G.B0.shape) == (d, G.N)      # stored as a diagonal
        for _ in range(2):
            S.engine.generator_step(G)
            S.engine.generator_step(G)
            S.engine.discriminator_step(G)
        if graphs:
            assert all(v is not False for v in G.graphs.values()) and len(G.graphs) >= 2, G.graphs    # every segment captured
        outs.append((S.engine.theta.data.clone(), S.engine.phi.data.clone(), S.engine.scal.clone()))
    close(outs[0][0], outs[1][0], 1e-10, 1e-13, 'theta: graph replay == eager')
    close(outs[0][1], outs[1][1], 1e-10, 1e-13, 'phi')


def test_structure_guard_catches_region_dependent_coefficients():
    """a coefficient that looks like the identity on the probe points but deviates elsewhere must not silently take the
    fused fast path: the guard on the actual sample raises"""
    from xnode_wan_pde_solver_amd._lib import XnwanError
    from xnode_wan_pde_solver_amd.engine import Structure
    z_params = {'alpha': 1e3, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
                'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
                'dim': 3, 'N_t': 5, 'N_r': 400, 'N_b': 60, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
                'domain': 'Hypercube'}

    def fa(X, i, j):                          # identity except in a slab none of the 18 (fixed-seed) probe points lies in
        base = torch.ones(X.shape[:-1]) if i == j else torch.zeros(X.shape[:-1])
        return base.to(X.device) + (1.0 if i == j else 0.0) * ((X[..., 1] > 0.66) & (X[..., 1] < 0.80)).float()

    funcs = dict(FUNCS, a=fa)
    st = Structure(funcs, 3)
    assert st.a_identity                      # the probe is fooled ...
    from src.training import NODE_WAN_solver
    torch.manual_seed(5)
    S = NODE_WAN_solver(z_params, fa, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
    domain, pts = first_sample(S)
    with pytest.raises(XnwanError):
        S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)


@pytest.mark.parametrize('Hh,Kk,Ww,m,q', [(16, 8, 32, 8, 9), (24, 9, 40, 5, 4), (20, 12, 50, 8, 9), (32, 12, 50, 3, 9),
                                          (20, 10, 50, 1, 9), (7, 3, 11, 2, 2), (20, 10, 64, 8, 9), (16, 8, 57, 3, 5),
                                          (20, 10, 50, 10, 9), (30, 11, 50, 9, 9),          # (u_layers 9, 10: the deepest compiled fields)
                                          (48, 16, 100, 8, 9), (64, 16, 128, 3, 4), (33, 10, 50, 8, 9), (20, 10, 70, 2, 9),
                                          (40, 14, 96, 9, 3), (48, 16, 64, 10, 4), (20, 10, 50, 12, 3)])
def test_engine_at_other_network_widths(Hh, Kk, Ww, m, q):
    """src/model.py:30-43,62-85,130-138 accept any u_hidden_dim, u_hidden_hidden_dim, v_hidden_dim and u_layers >= 1.  A
    network narrower than a kernel instantiation runs EXACTLY inside the next larger one (zero-padded blob, nets.Blob):
    same initial draws as the reference's construction order, sub-steps against the oracle at the true widths, the
    padding still identically zero after the updates, state_dict with the true shapes.  From (48, 16, 100) on the cases are wider
    than the round-5 containers: since round 6 they run inside the wide ones ((64, 16) for the stepper, 96 / 128 for the test
    network); (20, 10, 70) mixes the narrow stepper with the 96-wide test network, (48, 16) x 10 layers is the wide container
    at its deepest (two ReLU-mask words per stage), and the last case -- u_layers = 12, deeper than any container -- is what still
    runs at its own widths on the generic path (csrc/xw_generic.hip, u_layers <= 32), mixed with the MFMA kernels of the test
    network."""
    from oracle import refspec as R
    from src.training import NODE_WAN_solver
    from xnode_wan_pde_solver_amd import kernels as KN
    d = 4
    params = {'alpha': 1e3, 'u_layers': m, 'u_hidden_dim': Hh, 'u_hidden_hidden_dim': Kk, 'v_layers': q, 'v_hidden_dim': Ww,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': d, 'N_t': 7, 'N_r': 75, 'N_b': 41, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
              'domain': 'Hypercube'}
    torch.manual_seed(9)
    S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), './',
                        func_u_sol=P.func_u_sol, p=2)
    assert (S.engine.H, S.engine.K) == KN.ode_container(Hh, Kk, m) and S.engine.W == KN.disc_container(Ww)
    assert S.engine.generic == ((Hh, Kk, m) == (20, 10, 12), False)
    torch.manual_seed(9)
    O = R.Solver(params, FUNCS, u_sol=P.func_u_sol, p=2)
    # (u_layers = 1: a field without the tied hidden layer, src/model.py:130 -- no such parameters in the module)
    unames = [(n_, k_) for n_, k_ in R.u_names(m) if m > 1 or k_ not in ('Wh', 'Wh_b')]
    for n_, k_ in unames:
        assert torch.equal(dict(S.u_net.named_parameters())[n_].detach().cpu(), O.theta[k_]), n_
    for n_, k_ in R.V_NAME_MAP:
        assert torch.equal(dict(S.v_net.named_parameters())[n_].detach().cpu(), O.phi[k_]), n_
    sd = S.u_net.state_dict()
    assert sd['module.final_linear.weight'].shape == (1, Hh) and sd['module.initial_layers.2.weight'].shape == (Hh, Hh)
    rng = torch.get_rng_state()
    domain, pts = first_sample(S)
    torch.set_rng_state(rng)
    O.new_sample()
    G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
    for step in ('u', 'u', 'v', 'u'):
        if step == 'u':
            o = O.generator_step()
            S.engine.generator_step(G)
            got, blob, names = S.engine.grad_u, S.engine.theta, unames
            close(S.engine.scal[4], o['loss'], 1e-7)           # (the oracle rounds nabla u, nabla phi to float32 like the reference)
        else:
            o = O.discriminator_step()
            S.engine.discriminator_step(G)
            got, blob, names = S.engine.grad_v, S.engine.phi, R.V_NAME_MAP
            close(S.engine.scal[5], o['loss'], 1e-6, 1e-6)     # (-int = a difference of two logs: can sit near zero)
        gmax = max(float(o['grad'][k].abs().max()) for _, k in names)
        for (n_, k_), g_ in zip(names, blob.split(got)):
            close(g_, o['grad'][k_], 1e-5, 1e-6 * gmax, 'grad ' + k_)
    for (n_, k_), p_ in zip(unames, S.engine.theta.params):
        close(p_, O.theta[k_], 1e-6, 1e-8, 'theta after the updates: ' + k_)
    # the padding of the container blobs never left zero (weights, Adam moments): mask of the real entries
    for blob, st in ((S.engine.theta, S.engine.adam_u), (S.engine.phi, S.engine.adam_v)):
        real = torch.zeros_like(blob.data, dtype=torch.bool)
        for piece in blob.split(real):
            piece.fill_(True)
        assert int(real.sum()) == sum(blob.sizes)
        pad = ~real
        assert float(blob.data[pad].abs().sum()) == 0.0 and float(st['m'][pad].abs().sum()) == 0.0 and float(st['v'][pad].abs().sum()) == 0.0
    # module path at these widths
    with torch.no_grad():
        out = S.u_net(pts.interioru)
    close(out.squeeze(2), R.u_net(O.theta, params, O.X, P.func_h(O.X[:, 0, :])), 1e-5, 1e-7)     # (parameters agree to 1e-6 after four updates)


def test_custom_operators_pass_opcheck(golden_dir):
    """the registered operators (ops.py) against torch.library.opcheck's schema / fake-tensor / dispatch checks on real inputs"""
    z, params = load(golden_dir, 'ref_tiny_midpoint')
    S = make_solver(params, int(z['seed']))
    domain, pts = first_sample(S)
    X = pts.interioru.detach().cuda()
    net = S.u_net.module
    start = net.start_values(pts.interioru.detach()).cuda()
    args = (X, start, net.blob.data, net.method, net.kdims[0], net.kdims[1], net.num_layers, True)
    torch.library.opcheck(torch.ops.xnwan.xnode_forward, args, test_utils=('test_schema', 'test_faketensor'))
    u, Y = torch.ops.xnwan.xnode_forward(*args)
    close(u.squeeze(2), z['gen1/u'], F32TOL, F32TOL)
    torch.library.opcheck(torch.ops.xnwan.xnode_backward, (torch.ones_like(u), X, start, Y, net.blob.data, net.method, net.kdims[0],
                                                           net.kdims[1], net.num_layers, False, True),
                          test_utils=('test_schema', 'test_faketensor'))
    vn = S.v_net.module
    XV = pts.interiorv.detach().cuda()
    torch.library.opcheck(torch.ops.xnwan.testnet_forward, (XV, vn.blob.data, vn.kwidth, vn.num_layers), test_utils=('test_schema', 'test_faketensor'))
    v = torch.ops.xnwan.testnet_forward(XV, vn.blob.data, vn.kwidth, vn.num_layers)
    close(v.squeeze(2), z['gen1/v'], 1e-10, 1e-12)
    torch.library.opcheck(torch.ops.xnwan.testnet_backward, (torch.ones_like(v), XV, vn.blob.data, vn.kwidth, vn.num_layers, True, True),
                          test_utils=('test_schema', 'test_faketensor'))


def test_pipelined_training_loop_leaves_exactly_what_the_synchronous_one_does(tmp_path):
    """solver.pipeline: losses, diagnostic and best weights of iteration k are read back and written while iteration k + 1
    runs -- the parameters, the loss list, the files and the best weights must be those of the synchronous loop, bit for bit"""
    import hashlib
    params = {'alpha': 1e8, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 4, 'N_t': 9, 'N_r': 200, 'N_b': 100, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 12,
              'domain': 'Hypercube'}
    out = []
    cwd = os.getcwd()
    seen = []

    def never(solver, pts, domain):
        """the configs' acceptance rule (what the reference's main.py passes as `stop`), evaluated after every generator
        sub-iteration and never taken: the hook is handed ONE device tensor per sample and u_net(that tensor) is served from the
        loaded group (solver._stop_agreed) -- the same error the module path computes on the loader's own path tensor"""
        from utils.auxillary_funcs import rel_err
        err = float(rel_err(pts, solver.u_net, solver.func_u_sol, solver.p, domain.V(), solver.params['N_r']))
        seen.append((err, pts.is_cuda, pts))
        return err < 0.0

    for pipe in (True, False, 'hook'):
        wd = tmp_path / ('pipe_%s' % pipe)
        wd.mkdir()
        os.chdir(wd)
        try:
            S = make_solver(params, 3)
            S.pipeline = pipe is True
            S.capture_refill = pipe is not False   # (the plain synchronous run also refills by eager load_group calls and evaluates the diagnostic eagerly)
            if pipe == 'hook':
                S.stop = never
            losses = list(S.train(report=False))
            torch.cuda.synchronize()
            best = torch.load('best_model_weights_NODE.pth')
            out.append((losses, S.engine.theta.data.cpu(), S.engine.phi.data.cpu(), open('losses_NODE_4.json').read(),
                        open('L2_NODE_4.json').read(), hashlib.sha1(b''.join(v.cpu().numpy().tobytes() for v in best.values())).hexdigest(),
                        list(best.keys()), S.best_l))
        finally:
            os.chdir(cwd)
    a, b, c = out
    for other in (b, c):
        assert len(a[0]) >= 12 and a[0] == other[0] and torch.equal(a[1], other[1]) and torch.equal(a[2], other[2])
        assert a[3] == other[3] and a[4] == other[4] and a[5] == other[5] and a[6] == other[6] and a[7] == other[7]
    # the hook ran after every generator sub-iteration, on one device tensor per sample, and saw the error of the module path
    assert len(seen) == 24 and all(s_[1] for s_ in seen) and all(seen[2 * k][2] is seen[2 * k + 1][2] for k in range(12))
    assert all(seen[2 * k][2] is not seen[2 * k + 2][2] for k in range(11))
    X_last = seen[-1][2]
    from utils.auxillary_funcs import rel_err
    np.testing.assert_allclose(seen[-1][0], float(rel_err(X_last.clone(), S.u_net, S.func_u_sol, S.p, S._new_domain().V(), 200)),
                               rtol=1e-9)


def test_stop_hook_taken_leaves_what_the_reference_leaves(golden_dir, tmp_path):
    """the reference's own train() with a hook that fires at its fifth call (tests/golden/make_golden.py stop_taken): the loss
    list on disk, the number of diagnostics / times written before it left, the weights it saved under <path> and the best
    weights in the working directory -- against this engine's run with the same seed"""
    z, params = load(golden_dir, 'ref_stop_taken_d3_seed15')
    fire_at, calls = int(z['fire_at']), []

    def hook(solver, pts, dom):
        calls.append(1)
        return len(calls) == fire_at
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        out = str(tmp_path) + os.sep + 'run_'
        from src.training import NODE_WAN_solver
        torch.manual_seed(int(z['seed']))
        np.random.seed(int(z['seed']))
        S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), out,
                            stop=hook, func_u_sol=P.func_u_sol, p=2)
        S.tabulate_on_host = True
        with pytest.raises(SystemExit):
            S.train(report=False)
        d = params['dim']
        losses = json.load(open('losses_NODE_%d.json' % d))
        close(losses, z['losses'], 2e-5, what='loss list')
        assert len(json.load(open('Time_NODE_%d.json' % d))) == int(z['n_times']) and len(json.load(open('L2_NODE_%d.json' % d))) == int(z['n_L2'])
        saved, best = torch.load(out + 'best_model_weights_NODE.pth'), torch.load('best_model_weights_NODE.pth')
        assert list(saved.keys()) == [str(k) for k in z['saved_keys']] == list(best.keys())
        for k_ in saved:
            close(saved[k_], z['saved/' + k_], 2e-5, 2e-6, 'saved ' + k_)
            close(best[k_], z['best/' + k_], 2e-5, 2e-6, 'best ' + k_)
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize('domain', ['Hypercube', 'NSphere_TCone'])
def test_stop_hook_taken_saves_the_weights_of_that_moment_and_leaves(tmp_path, domain):
    """src/training.py:142-146: the moment `stop` returns True the generator's weights go to <path>best_model_weights_NODE.pth
    and the process exits (solver.exit_on_stop = False: train() returns instead).  Cube (captured sub-steps, served hook
    tensor) and a ball domain (group runner): the loss list ends with the sub-iteration whose hook fired, the file holds the
    parameters of that moment -- one generator sub-iteration after the previous one, no discriminator sub-step in between --
    and nothing of the interrupted outer iteration's later files exists."""
    params = {'alpha': 1e6, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 3, 'N_t': 7, 'N_r': 150, 'N_b': 90, 'T0': 0, 'T': 1, 'iterations': 6, 'domain': domain,
              'shape_param': [-1, 1] if domain == 'Hypercube' else 1.0}
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        calls = []

        def fifth(solver, pts, dom):
            calls.append(solver.engine.theta.data.clone())
            return len(calls) == 5                      # (outer iteration 2, first generator sub-iteration)

        out = str(tmp_path) + os.sep + 'run_'
        from src.training import NODE_WAN_solver
        torch.manual_seed(2)
        np.random.seed(2)
        S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), out,
                            stop=fifth, func_u_sol=P.func_u_sol, p=2)
        S.exit_on_stop = False
        losses = list(S.train(report=False))
        torch.cuda.synchronize()
        assert len(calls) == 5 and len(losses) == 5 and all(np.isfinite(losses))
        saved = torch.load(out + 'best_model_weights_NODE.pth')
        assert list(saved.keys()) == list(S.u_net.state_dict().keys())
        for k_, v_ in S.u_net.state_dict().items():
            assert torch.equal(saved[k_].cpu(), v_.cpu()), k_
        assert torch.equal(S.engine.theta.data, calls[4]) and not torch.equal(calls[4], calls[3])
        assert json.load(open('losses_NODE_3.json')) == losses
        assert len(json.load(open('Time_NODE_3.json'))) == 3         # (start + two complete outer iterations before the one that stopped)
        # the reference leaves the process: the default does too
        calls.clear()
        torch.manual_seed(2)
        np.random.seed(2)
        S2 = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cuda'), out,
                             stop=fifth, func_u_sol=P.func_u_sol, p=2)
        with pytest.raises(SystemExit):
            S2.train(report=False)
        assert len(calls) == 5 and torch.equal(calls[4], S.engine.theta.data)
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize('case', ['ref_traj_cone_ex43_d3_seed0', 'ref_traj_hourglass_ex43_d3_seed1',
                                  'ref_traj_cone_ex43_d10_full_seed2', 'ref_traj_hourglass_ex43_d10_full_seed3',
                                  'ref_traj_hourglass_ex43_d3_euler_seed7', 'ref_traj_cone_ex43_d3_rk4_seed8'])
def test_ball_domain_fast_loop_ends_where_the_reference_ends(golden_dir, tmp_path, case):
    """The same reference runs as above through the loop a hook-free train() takes on the ball domains -- one read-back per outer
    iteration, the next sample loaded behind the queued sub-steps, the samples drawn by the forked sampling process, h / f / g
    tabulated on the GPU -- compared where that loop can be compared without a hook: u_theta on the fixture's probe after the
    last outer iteration against the reference's last logged value (theta does not change after the last generator
    sub-iteration), and the loss list's length.  The d = 3 cases are 100 / 60 outer iterations of ~12 groups each."""
    import configs.Ex4_3_funcs as F
    z, params = load(golden_dir, case)
    params.pop('funcs')
    ref = z['rel_l2']
    probe, sol = torch.from_numpy(z['probe']), torch.from_numpy(z['probe_sol'])
    S = make_solver(params, int(z['seed']), F=F)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        losses = S.train(report=False)
    finally:
        os.chdir(cwd)
    assert '_list_phase_seconds' in S.__dict__                                     # (the loop under test ran ...
    assert getattr(S, '_sampler_proc', None) is not None or not S.sampler_process   #  ... with the sampling process, unless this machine cannot start one: then it said so and drew on the helper thread)
    assert len(losses) == len(ref)
    with torch.no_grad():
        up = S.u_net(probe).squeeze(2).cpu()
    got = float(torch.sqrt(torch.mean((up - sol) ** 2) / torch.mean(sol ** 2)))
    out = os.environ.get('XW_DUMP_TRAJ')
    if out:
        print('%s: probe error after the last outer iteration %.12g, reference %.12g (relative difference %.2e)'
              % (case, got, float(ref[-1]), abs(got - float(ref[-1])) / float(ref[-1])))
    # (float64 samples end to end; measured on the MI355X: 1.5e-10 relative after 100 outer iterations on the cone, 7e-14 at full size)
    np.testing.assert_allclose(got, float(ref[-1]), rtol=1e-7)
    assert abs(got - float(ref[-1])) < 1e-2


@pytest.mark.parametrize('domain,seed', [('NSphere_THourglass', 8), ('NSphere_TCone', 9)])
def test_all_groups_of_a_list_sample_loaded_by_one_gather_launch_hold_what_load_group_writes(domain, seed):
    """Engine.load_groups_packed (the sample fields of every group as regions of the group's one allocation, filled by ONE
    xw_gather_fields launch from the uploaded sample and the whole-sample tables of tabulate_sample) against load_group, group by
    group: every sample field bit for bit, the same shapes, flags and pair terms -- late-entry groups of the hourglass (per-point
    times), single-slice groups at T0 (the factorised [N, N] terms) and boundary groups included -- and one outer iteration on
    top gives the same parameters either way"""
    import configs.Ex4_3_funcs as F
    from xnode_wan_pde_solver_amd.engine import Group
    params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 5, 'N_t': 9, 'N_r': 1500, 'N_b': 700, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 1, 'domain': domain}
    out = []
    for packed in (True, False):
        S = make_solver(params, seed, F=F)
        S.engine.packed_load = packed
        dom = S._new_domain()
        pts = S._loader(dom).pin()
        groups = S._prepare_groups(pts, dom)
        assert len(groups) > 3 and (('_tab_cat' in S.engine.__dict__) and all('xT' in G._lazy for G in groups)) == packed
        fields = []
        for G in groups:
            rec = {k: (None if getattr(G, k, None) is None else getattr(G, k).clone()) for k in Group.SAMPLE_FIELDS}
            rec['meta'] = (G.N, G.L, G.Nb, G.Lb, G.same_grid, G.Vol, G.Nglob, G.Nbglob, G.pair_i, G.pair_b, G.init_off, G.bdry_off, G.s3_scale, G.amode)
            fields.append(rec)
        for G in groups:
            G.persistent = False
        S.engine.begin_substep('u', True)
        for G in groups:
            S.engine.generator_step(G)
        S.engine.begin_substep('v', True)
        for G in groups:
            S.engine.discriminator_step(G)
        out.append((fields, S.engine.theta.data.clone(), S.engine.phi.data.clone()))
    a, b = out
    assert len(a[0]) == len(b[0])
    kinds = set()
    for ra, rb in zip(a[0], b[0]):
        assert ra['meta'] == rb['meta']
        kinds.add((ra['tpp'] is not None, ra['meta'][8], ra['meta'][9]))
        for k in Group.SAMPLE_FIELDS:
            assert (ra[k] is None) == (rb[k] is None), k
            if ra[k] is not None:
                assert ra[k].shape == rb[k].shape and ra[k].dtype == rb[k].dtype and torch.equal(ra[k], rb[k]), k
    assert (False, False, True) in kinds or (False, True, True) in kinds          # a single-slice boundary group at T0
    if domain == 'NSphere_THourglass':
        assert any(k_[0] for k_ in kinds)                                         # late-entry groups with per-point times
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize('domain,seed', [('NSphere_THourglass', 5), ('NSphere_TCone', 6)])
def test_list_domain_loop_with_one_read_back_leaves_exactly_what_the_synchronous_one_does(tmp_path, domain, seed):
    """solver.defer_list_readback (ball domains, 11-20 groups per sample): all sub-steps of an outer iteration queued without a
    read-back, the next sample's groups loaded behind them, losses / theta snapshots / diagnostic read back once -- with the
    samples drawn by the forked sampling process (solver.sampler_process, sampler_proc.py) or by the helper thread.  The
    parameters, the loss list, the files, the best weights and the generator streams after train() must be those of the loop
    that synchronises after every sub-iteration (the one the reference-trajectory fixtures run through), bit for bit."""
    import hashlib
    import configs.Ex4_3_funcs as F
    params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 4, 'N_t': 9, 'N_r': 600, 'N_b': 400, 'T0': 0, 'T': 1, 'shape_param': 1.0, 'iterations': 7, 'domain': domain}
    out = []
    cwd = os.getcwd()
    for mode, (defer, proc) in enumerate(((True, True), (True, False), (False, False))):
        wd = tmp_path / ('mode%d' % mode)
        wd.mkdir()
        os.chdir(wd)
        try:
            S = make_solver(params, seed, F=F)
            S.defer_list_readback, S.sampler_process = defer, proc
            S.engine.packed_load = proc          # (... and the groups' fields from one gather launch / group by group)
            losses = list(S.train(report=False))
            S.iterations = 3
            losses += list(S.train(report=False))          # (a second call: the sampling process is handed the streams again)
            torch.cuda.synchronize()
            assert len(S._group_cache) > 1
            assert (getattr(S, '_sampler_proc', None) is not None) == (proc and S.sampler_process)   # (sampler_process goes False, with a warning, where no child can be started)
            best = torch.load('best_model_weights_NODE.pth')
            out.append((losses, S.engine.theta.data.cpu(), S.engine.phi.data.cpu(), open('losses_NODE_4.json').read(),
                        open('L2_NODE_4.json').read(), hashlib.sha1(b''.join(v.cpu().numpy().tobytes() for v in best.values())).hexdigest(),
                        list(best.keys()), S.best_l, S.last_loss_u, S.last_loss_v, len(json.load(open('Time_NODE_4.json'))),
                        torch.rand(4).tolist(), np.random.rand(4).tolist(), np.random.normal(size=3).tolist()))
            if getattr(S, '_sampler_proc', None) is not None:
                S._sampler_proc[1].close()
        finally:
            os.chdir(cwd)
    a, b, c = out
    assert len(a[0]) == 20
    for other in (b, c):
        assert a[0] == other[0] and torch.equal(a[1], other[1]) and torch.equal(a[2], other[2])
        assert a[3:] == other[3:]


def test_poisoned_work_buffers_change_nothing(golden_dir, monkeypatch):
    """XW_POISON=1 starts every work buffer as NaN: no kernel may read a slot that nobody wrote (a stale but plausible value
    from the allocator is how such a read hides), and nothing may depend on timing -- the same three sub-steps, bit for bit"""
    z, params = load(golden_dir, 'ref_d20_small_midpoint')
    outs = []
    from xnode_wan_pde_solver_amd.options import EngineOptions
    for poison in ('0', '1'):
        S = make_solver(params, int(z['seed']), options=EngineOptions(poison=poison == '1'))
        domain, pts = first_sample(S)
        G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        if poison == '1':
            assert torch.isnan(G.Y).all() and torch.isnan(G.act).all()
        S.engine.generator_step(G)
        S.engine.generator_step(G)
        S.engine.discriminator_step(G)
        outs.append((S.engine.theta.data.clone(), S.engine.phi.data.clone(), S.engine.scal.clone()))
    assert all(torch.isfinite(t_).all() for t_ in outs[1])
    assert all(torch.equal(a_, b_) for a_, b_ in zip(*outs))


def test_solvers_that_go_out_of_scope_do_not_take_live_graphs_with_them(golden_dir):
    """Engine keeps every captured sub-step graph alive (engine._KEPT_GRAPHS): on this stack destroying a multi-branch HIP
    graph can make a later launch of ANOTHER live graph fault inside the runtime (seen once, in a sequence of 27 tests; this
    short sequence does not provoke it on its own).  What is checked here: a solver is trained, others are created, trained and
    collected around it, their graphs stay registered, and the first one goes on replaying its graphs with results identical to
    an undisturbed twin."""
    import gc
    from xnode_wan_pde_solver_amd import engine as E
    z, params = load(golden_dir, 'ref_plumb_midpoint')

    def fresh():
        S = make_solver(params, int(z['seed']))
        domain, pts = first_sample(S)
        G = S.engine.load_group(pts.interioru, pts.interiorv, pts.boundary, domain)
        return S, G

    def cycle(S, G, n):
        for _ in range(n):
            S.engine.generator_step(G); S.engine.generator_step(G); S.engine.discriminator_step(G)
    A, GA = fresh()
    T, GT = fresh()                      # the undisturbed twin
    cycle(A, GA, 2); cycle(T, GT, 2)
    kept = len(E._KEPT_GRAPHS)
    for k in range(6):
        B, GB = fresh()
        cycle(B, GB, 2)
        del B, GB
        gc.collect()
        cycle(A, GA, 2); cycle(T, GT, 2)
    torch.cuda.synchronize()
    assert len(E._KEPT_GRAPHS) > kept                            # the collected solvers' graphs are still there
    assert torch.equal(A.engine.theta.data, T.engine.theta.data) and torch.equal(A.engine.phi.data, T.engine.phi.data)
    assert torch.isfinite(A.engine.theta.data).all()


@pytest.mark.parametrize('general', [False, True])
def test_group_refilled_by_one_graph_replay_holds_what_load_group_writes(general):
    """Engine.refill_compact (the training loop's refill of its group: static input buffers + ONE captured graph, with the
    lean field-by-field body where the coefficient structure is the fused one) against load_group on the same samples, every
    sample field bit for bit, on the capture pass and on plain replays; a general b_i takes load_group's own body (captured
    when its tabulation allows it, eager with a warning otherwise) -- same fields."""
    import warnings
    from xnode_wan_pde_solver_amd import sampling
    from xnode_wan_pde_solver_amd.engine import Group
    params = {'alpha': 1e8, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 5, 'N_t': 9, 'N_r': 300, 'N_b': 120, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
              'domain': 'Hypercube'}

    class F(object):
        pass
    for k in ('func_a', 'func_c', 'func_h', 'func_f', 'func_g', 'func_u_sol'):
        setattr(F, k, staticmethod(getattr(P, k)))
    F.func_b = staticmethod((lambda X, i: 0.1 * (i + 1) * X[..., 1]) if general else P.func_b)
    S = make_solver(params, 5, F=F)
    eng, dev = S.engine, S.device
    samples, refs = [], []
    for _ in range(3):
        dom, pts = first_sample(S)
        comp = pts.compact()
        td = comp[0].to(dev)
        X, XV, BX = (sampling._paths(td, c.to(dev)) for c in comp[1:])
        samples.append((dom, comp))
        refs.append(eng.load_group(X, XV, BX, dom, shared_grid_t0=float(comp[0][0])))
    G = refs.pop()                       # the group that gets refilled; the other two are what it must hold afterwards
    samples.pop()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        for rep in range(4):
            dom, comp = samples[rep % 2]
            ver = G.sample_version
            assert eng.refill_compact(G, comp, dom) is G and G.sample_version == ver + 1 and G.domain is dom
            for k in Group.SAMPLE_FIELDS:
                a, b = getattr(G, k), getattr(refs[rep % 2], k)
                assert (a is None) == (b is None), k
                assert a is None or torch.equal(a, b), (k, rep)
    kinds = [type(v).__name__ for k, v in G.graphs.items() if k.startswith('refill')]
    assert kinds in ((['bool'], ['CUDAGraph']) if general else (['CUDAGraph'],)), kinds


def test_plan_names_the_loop_a_solver_takes():
    """NODE_WAN_solver.plan(): the default-path matrix (domain x stop hook x report x switches) made explicit"""
    params = {'alpha': 1e4, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
              'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
              'dim': 4, 'N_t': 8, 'N_r': 64, 'N_b': 32, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1, 'domain': 'Hypercube'}
    S = make_solver(params, 0)
    p = S.plan()
    assert p['loop'].startswith('pipelined') and p['sampling'] == 'helper thread' and p['refill'].startswith('one graph replay')
    assert p['sub_steps'] == 'captured HIP graphs' and p['ranks'] == 1 and p['exchange'] is None and p['coefficients'].startswith('a=identity')
    assert S.plan(report=True)['loop'].startswith('synchronous')
    S.stop = lambda *a: False
    p = S.plan()
    assert p['loop'].startswith('synchronous') and p['sampling'].startswith('in the loop')
    C = make_solver(dict(params, domain='NSphere_TCone', shape_param=1.0), 0)
    p = C.plan()
    assert p['loop'].startswith('list domain, one read-back') and p['sampling'] == 'forked sampling process'
    assert p['refill'].startswith('one packed upload') and p['sub_steps'].startswith('one C call per group sub-step')
    C.sampler_process = False
    assert C.plan()['sampling'] == 'helper thread'
    C.defer_list_readback = False
    assert C.plan()['loop'].startswith('synchronous')
