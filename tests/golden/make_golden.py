#!/usr/bin/env python3
"""Generate golden vectors by running the upstream reference IN THIS CONTAINER ONLY.

Usage:  python tests/golden/make_golden.py [--traj]          (needs /root/reference; CPU; ~1 min, --traj ~10 min)
Round-4 fixtures, one option each (reference CPU time on this container's 8 cores):
        --round4                  ref_d20_headline (one outer iteration at the benchmarked size, slim record) + the r = 0.7 samplers   4 min
        --traj-headline           ref_traj_d20_headline_seed4: 250 outer iterations of the reference's train() at the benchmarked size  25 min
        --traj-headline-solvers   ref_traj_d20_headline_{euler_seed5, rk4_seed6}: 30 outer iterations each                              8 min
        --traj-cfg5               ref_traj_{cone, hourglass}_ex43_d10_full_*: 8 outer iterations at config 5's stated size              4 min
        --traj-ball-solvers       ref_traj_hourglass_ex43_d3_euler_seed7, ref_traj_cone_ex43_d3_rk4_seed8: 40 outer iterations each    2 min
        --general                 ref_general_d4_midpoint: general a_ij, b_i, c(u,t,x) (general_funcs.py)                               seconds
        --shapes                  ref_wide_d6_midpoint, ref_narrow_d3_euler, ref_m1_d4_rk4: other network shapes                        seconds
        --proj                    ref_proj: what the reference's proj() saves (three slices); ref_stop_taken_d3_seed15: a stop hook that fires                                           seconds
        --alpha1                  ref_alpha1_*: alpha = 1 (cube midpoint / rk4 / general euler; cone, hourglass, hourglass general)                 seconds
        --intervals               ref_interval_d4_midpoint, ref_interval_d3_rk4: other time intervals and cubes                              seconds
        --loops                   ref_traj_n1_3_n2_2_d3_seed16, ref_traj_n1_1_n2_3_d3_seed17, ref_traj_cone_n1_3_n2_2_d3_seed18: other (n1, n2)         seconds
        --generic                 ref_generic_d5_midpoint, ref_generic_d3_rk4, ref_generic_mixed_d4_euler: widths of the generic path   seconds
(general b_i: the reference's `np.sum(list of tensors)` goes through shim 2 below, i.e. Python's sum over the list.)

What this is: test infrastructure.  It imports the reference implementation from
/root/reference (never copied into this repo, never shipped to the GPU box) and
records inputs/outputs of its hot path as small .npz fixtures under tests/golden/.
Those fixtures pin the oracle (oracle/refspec.py) and, through it, the HIP path.

Two harness-level shims are needed to import the reference on this software stack
(SURVEY.md Appendix C):
  1. `torchdiffeq` (requirements.txt:7, imported at src/model.py:8) is a third-party
     dependency that is not installed and not vendored under /root/reference.  A
     stand-in package is written to a temp dir: fixed grid == requested t, t cast to
     y0's dtype, euler / midpoint / 3/8-rule rk4.  Because the stepper arithmetic
     lives in that third-party package, fixtures that depend on it pin everything
     EXCEPT the stepper definition itself ("parity unpinned" for the stepper; see
     DESIGN.md §Oracle).
  2. src/loss.py:69 calls np.sum on a list of grad-requiring tensors, which raises on
     numpy>=2; the name `np` inside the loaded src.loss module is replaced by an
     object whose `sum` is Python's builtin sum.  With b == 0 (all shipped PDEs) the
     term is exactly zero either way.

The sequence of calls inside `one_iteration` follows NODE_WAN_solver.train
(src/training.py:118-162) for the first outer iteration; sub-steps after the first
use `.clone()`d sample tensors, which is what `.to(device)` produces on a GPU
(src/dataset.py:321), i.e. the semantics the HIP engine implements.
"""
import argparse
import builtins
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

_TORCHDIFFEQ_STANDIN = '''
import torch
def _incr(method, f, t, dt, y):
    if method == 'euler':
        return dt * f(t, y)
    if method == 'midpoint':
        return dt * f(t + dt / 2, y + f(t, y) * dt / 2)
    if method == 'rk4':
        k1 = f(t, y)
        k2 = f(t + dt / 3, y + dt * k1 / 3)
        k3 = f(t + 2 * dt / 3, y + dt * (k2 - k1 / 3))
        k4 = f(t + dt, y + dt * (k1 - k2 + k3))
        return dt * (k1 + 3 * (k2 + k3) + k4) / 8
    raise ValueError(method)
def odeint(func, y0, t, method='midpoint', **unused):
    t = t.type_as(y0)
    out = [y0]
    y = y0
    for t0, t1 in zip(t[:-1], t[1:]):
        y = y + _incr(method, func, t0, t1 - t0, y)
        out.append(y)
    return torch.stack(out, 0)
odeint_adjoint = odeint
'''


def load_reference(funcs_module='configs.Ex4_1_funcs', dim=None):
    """funcs_module = 'configs.Ex4_3_funcs' needs a third harness-level shim: that file reads
    the dimension from `NODE_GAN.main.params` (configs/Ex4_3_funcs.py:3), a module that does not exist in the reference
    tree; a stand-in module carrying only {'dim': dim} is registered before the import."""
    if 'src.training' in sys.modules and getattr(load_reference, 'ready', False):
        return _reference_modules(funcs_module, dim)
    shim = tempfile.mkdtemp(prefix='tdq_standin_')
    os.makedirs(os.path.join(shim, 'torchdiffeq'))
    with open(os.path.join(shim, 'torchdiffeq', '__init__.py'), 'w') as fh:
        fh.write(_TORCHDIFFEQ_STANDIN)
    sys.path.insert(0, shim)
    sys.path.insert(1, REF)
    import matplotlib
    matplotlib.use('Agg')
    import src  # noqa: F401  (runs the reference's src/__init__.py)

    class _NP:
        sum = staticmethod(builtins.sum)
    sys.modules['src.loss'].np = _NP
    load_reference.ready = True
    return _reference_modules(funcs_module, dim)


def _reference_modules(funcs_module, dim):
    import importlib
    import types
    if funcs_module.endswith('Ex4_3_funcs'):
        if 'NODE_GAN.main' not in sys.modules:
            pkg, main = types.ModuleType('NODE_GAN'), types.ModuleType('NODE_GAN.main')
            main.params = {}
            pkg.main = main
            sys.modules['NODE_GAN'], sys.modules['NODE_GAN.main'] = pkg, main
        sys.modules['NODE_GAN.main'].params['dim'] = dim      # (the functions read params['dim'] at call time)
    funcs = importlib.import_module(funcs_module)
    return sys.modules['src.training'], sys.modules['src.dataset'], sys.modules['src.loss'], funcs


def make_params(d, N_r, N_b, N_t, solver='midpoint', iterations=1, alpha=100000000):
    # YAML key order (configs/cube_pde.yaml:2-23): 13 config, 7 setup, iterations, domain
    return {
        'alpha': alpha, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10,
        'v_layers': 9, 'v_hidden_dim': 50, 'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04,
        'min_steps': 5, 'adjoint': False, 'solver': solver,
        'dim': d, 'N_t': N_t, 'N_r': N_r, 'N_b': N_b, 'T0': 0, 'T': 1, 'shape_param': [-1, 1],
        'iterations': iterations, 'domain': 'Hypercube',
    }


def npy(t):
    return t.detach().cpu().numpy().copy()


def one_iteration(case, d, N_r, N_b, N_t, seed, solver_name, full_tensors, shape_param=None, slim=0, general=False, alpha=None,
                  net=None):
    """slim = s > 0 (round 4, the headline size N_r = 4096): per-path arrays keep every s-th path (`[::s]`; the dense
    float32 `dphi` every 4s-th), the whole samples are pinned by their SHA-1 instead; scalars, gradients and parameters
    are kept in full.  The consumer slices its own arrays the same way (`slim_stride` in the file)."""
    import hashlib
    training, dataset, lossmod, F = load_reference()
    if general:
        # round 4: the reference's own classes with GENERAL coefficient callables (tests/golden/general_funcs.py) in place of
        # Ex4_1's identity a, zero b, c = -u: pins src/training.py:32-41 (the a[d,d,N,L] / b[d,N,L] tables) and src/loss.py:66-69
        import types
        sys.path.insert(0, HERE)
        import general_funcs as GF
        fa, fb, fc = GF.variant('v1' if general is True else general)        # (round 5: general may name another set, 'const' / 'diag')
        F = types.SimpleNamespace(func_a=fa, func_b=fb, func_c=fc, func_h=F.func_h, func_f=F.func_f,
                                  func_g=F.func_g, func_u_sol=F.func_u_sol)
    sl = (lambda a: a[::slim].copy()) if slim else (lambda a: a)
    sl4 = (lambda a: a[::4 * slim].copy()) if slim else (lambda a: a)
    sha = lambda a: np.array(hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest())  # noqa: E731
    params = make_params(d, N_r, N_b, N_t, solver_name) if alpha is None else make_params(d, N_r, N_b, N_t, solver_name, alpha=alpha)
    if general:
        params['funcs'] = 'general_' + ('v1' if general is True else general)
    if net is not None:           # network shapes other than the YAML's (src/model.py:30-43,62-85,130-138 accept any)
        params.update(net)
    if shape_param is not None:
        # d = 100: with the YAML's integer [-1, 1] the reference's own diagnostic raises OverflowError (V() is the Python
        # int 2**100, utils/auxillary_funcs.py:15 multiplies a tensor by it); float bounds are what lets it run at all
        params['shape_param'] = shape_param
    dev = torch.device('cpu')
    torch.manual_seed(seed)
    np.random.seed(seed)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, dev, './',
                                 func_u_sol=F.func_u_sol, p=2)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(seed)}
    if slim:
        out['slim_stride'] = np.array(slim)
    # initial parameters: unique tensors by canonical name + the full state_dict key list with its aliasing
    # (tied layers appear under several keys, src/model.py:38,127-131; `module.` prefix from DataParallel)
    for tag, net in (('u', S.u_net), ('v', S.v_net)):
        canon = {}
        for n, p_ in net.named_parameters():
            out[tag + '_sd/' + n] = npy(p_)
            canon[p_.data_ptr()] = n
        sd = net.state_dict()
        out[tag + '_sd_keys'] = np.array(list(sd.keys()))
        out[tag + '_sd_alias_of'] = np.array([canon[v.data_ptr()] for v in sd.values()])
        out[tag + '_param_names'] = np.array([n for n, _ in net.named_parameters()])

    domain = S.domain(S.setup['shape_param'], d, S.setup['T0'], S.setup['T'], N_t)
    points = dataset.Comb_loader(N_r, N_b, domain, dev)
    out['times'] = npy(domain.times)
    for key, arr in (('x_u', points.interioru), ('x_v', points.interiorv), ('x_b', points.boundary)):
        out[key] = sl(npy(arr[:, 0, 1:]))
        if slim:
            out[key + '_sha1'] = sha(npy(arr[:, 0, 1:]))
    if full_tensors:
        out['X'] = npy(points.interioru)
        out['XV'] = npy(points.interiorv)
        out['BX'] = npy(points.boundary)
    out['V'] = np.array(float(domain.V()))
    out['w_v'] = sl(npy(domain.func_w(points.interiorv)))
    out['L2_start'] = np.array(training.L_norm(points.interioru, S.u_net, S.p, S.func_u_sol, domain.V(), N_r).item())
    out['rel_start'] = np.array(training.rel_err(points.interioru, S.u_net, S.func_u_sol, S.p, domain.V(), N_r).item())

    def net_step(tag, which, datau, datav, bdata, opt):
        """one optimiser sub-step exactly as in src/training.py:127-138 / :152-162"""
        opt.zero_grad()
        pv = S.v_net(datav)
        pu = S.u_net(datau)
        h, f, g, a, b, c = training.func_eval(datau.clone().detach(), bdata.clone().detach(), S.setup, pu,
                                              F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g)
        Lo = lossmod.loss(S.config['alpha'], a, b, c, h, f, g, S.setup, domain, dev)
        # side-effect-free views of what loss.I reads from X.grad / XV.grad (src/loss.py:55-63)
        G = torch.autograd.grad(pu.sum(), datau, retain_graph=True)[0]
        w = domain.func_w(datav).unsqueeze(2)
        dphi = torch.autograd.grad((pv * w).sum(), datav, retain_graph=True)[0]
        out[tag + '/u'] = sl(npy(pu.squeeze(2)))
        out[tag + '/v'] = sl(npy(pv.squeeze(2)))
        out[tag + '/h'] = sl(npy(h))
        out[tag + '/f'] = sl(npy(f))
        out[tag + '/g'] = sl(npy(g))
        out[tag + '/Xgrad_l0'] = sl(npy(G[:, 0, :]))
        out[tag + '/Xgrad_rest_absmax_x'] = np.array(float(G[:, 1:, 1:].abs().max()) if N_t > 1 else 0.0)
        out[tag + '/Xgrad_t_path0'] = npy(G[0, :, 0])
        out[tag + '/dphi'] = sl4(npy(dphi))
        if which == 'u':
            ub = S.u_net(bdata)
            out[tag + '/u_b'] = sl(npy(ub.squeeze(2)))
            out[tag + '/init'] = np.array(Lo.init(pu).item())
            out[tag + '/bdry'] = np.array(Lo.bdry(S.u_net, bdata).item())
            L = Lo.u(pu, pv, S.u_net, datau, datav, bdata)
        else:
            L = Lo.v(pu, pv, datau, datav)
        out[tag + '/loss'] = np.array(L.item())
        # I and int recomputed on a throw-away loss object would disturb .grad; derive from the loss value instead
        L.backward(retain_graph=True)
        net = S.u_net if which == 'u' else S.v_net
        for n, p in net.named_parameters():
            out[tag + '/grad/' + n] = npy(p.grad)
        opt.step()
        for n, p in net.named_parameters():
            out[tag + '/after/' + n] = npy(p)

    # sub-step 1 of the generator: the loader's own leaves (identical on CPU and GPU semantics)
    datau, datav, bdata = points[0]
    net_step('gen1', 'u', datau, datav, bdata, S.optimizer_u)
    # a separate evaluation of I/int for gen1, on clones so that no .grad state is shared
    fresh = lambda t: t.detach().clone().requires_grad_(True)  # noqa: E731
    # sub-step 2 and the discriminator step with GPU loader semantics (fresh differentiable copies per pass)
    datau2, datav2, bdata2 = fresh(points.interioru), fresh(points.interiorv), fresh(points.boundary)
    net_step('gen2', 'u', datau2, datav2, bdata2, S.optimizer_u)
    datau3, datav3, bdata3 = fresh(points.interioru), fresh(points.interiorv), fresh(points.boundary)
    net_step('disc1', 'v', datau3, datav3, bdata3, S.optimizer_v)

    # stand-alone I / int at the final parameters (pins src/loss.py:46-76,87-90 values)
    X4, XV4, B4 = fresh(points.interioru), fresh(points.interiorv), fresh(points.boundary)
    pv = S.v_net(XV4)
    pu = S.u_net(X4)
    h, f, g, a, b, c = training.func_eval(X4.clone().detach(), B4.clone().detach(), S.setup, pu,
                                          F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g)
    Lo = lossmod.loss(S.config['alpha'], a, b, c, h, f, g, S.setup, domain, dev)
    out['final/I'] = np.array(Lo.I(pu, pv, X4, XV4).item())
    X5, XV5 = fresh(points.interioru), fresh(points.interiorv)
    pv = S.v_net(XV5)
    pu = S.u_net(X5)
    c = F.func_c(X5.clone().detach(), pu)
    Lo = lossmod.loss(S.config['alpha'], a, b, c, h, f, g, S.setup, domain, dev)
    out['final/int'] = np.array(Lo.int(pu, pv, X5, XV5).item())
    out['final/u'] = sl(npy(pu.squeeze(2)))
    out['final/v'] = sl(npy(pv.squeeze(2)))

    # second sample of the iteration + diagnostic (src/training.py:166-167) pins the RNG stream position
    points2 = dataset.Comb_loader(N_r, N_b, domain, dev)
    out['x_u_second'] = sl(npy(points2.interioru[:, 0, 1:]))
    if slim:
        out['x_u_second_sha1'] = sha(npy(points2.interioru[:, 0, 1:]))
    out['L2_end'] = np.array(training.L_norm(points2.interioru, S.u_net, S.p, S.func_u_sol, domain.V(), N_r).item())
    path = os.path.join(HERE, case + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


def trajectory(case, d, N_r, N_b, N_t, seed, outer_iters, gpu_loader_semantics, solver_name='midpoint', net=None, general=False,
               hook_draws=False):
    """rel-L2 at every generator sub-step through the reference's own train() loop (stop hook = logging point)."""
    training, dataset, lossmod, F = load_reference()
    params = make_params(d, N_r, N_b, N_t, solver_name, iterations=outer_iters)
    if net is not None:
        params.update(net)
    if general:                                    # (round 5: tests/golden/general_funcs.py in place of Ex4_1's a, b, c)
        import types
        sys.path.insert(0, HERE)
        import general_funcs as GF
        F = types.SimpleNamespace(func_a=GF.func_a, func_b=GF.func_b, func_c=GF.func_c, func_h=F.func_h, func_f=F.func_f,
                                  func_g=F.func_g, func_u_sol=F.func_u_sol)
        params['funcs'] = 'general_v1'
    if gpu_loader_semantics:
        orig = dataset.Comb_loader.__getitem__

        def getitem(self, idx):
            r = orig(self, idx)
            return tuple(t.clone() for t in r)
        dataset.Comb_loader.__getitem__ = getitem
    log, losses = [], []

    def hook(self, pts, domain):
        with torch.no_grad():
            log.append(training.rel_err(pts, self.u_net, self.func_u_sol, self.p, domain.V(), self.params['N_r']).item())
        losses.append(self.av_l)
        if hook_draws:                             # (round 5: a hook that consumes both global generators between the sub-iterations)
            torch.rand(3)
            np.random.rand(2)
        return False
    torch.manual_seed(seed)
    np.random.seed(seed)
    cwd = os.getcwd()
    scratch = tempfile.mkdtemp(prefix='ref_traj_')
    os.chdir(scratch)
    try:
        S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g,
                                     torch.device('cpu'), './', func_u_sol=F.func_u_sol, p=2, stop=hook)
        t0 = time.time()
        S.train(report=False)
        wall = time.time() - t0
        # fixed probe set for the trained u
        g = torch.Generator().manual_seed(12345)
        bot, top = params['shape_param']
        xp = torch.rand(64, 1, d, generator=g) * 2 - 1
        tp = torch.linspace(0, 1, N_t).view(1, N_t, 1).repeat(64, 1, 1)
        if (bot, top, params['T0'], params['T']) != (-1, 1, 0, 1):          # (round 5: the probe lives on the run's own interval and cube)
            xp = bot + (top - bot) * (xp + 1) / 2
            tp = params['T0'] + (params['T'] - params['T0']) * tp
        probe = torch.cat((tp, xp.repeat(1, N_t, 1)), 2)
        with torch.no_grad():
            up = S.u_net(probe).squeeze(2)
    finally:
        os.chdir(cwd)
        if gpu_loader_semantics:
            dataset.Comb_loader.__getitem__ = orig
    path = os.path.join(HERE, case + '.npz')
    np.savez_compressed(path, rel_l2=np.array(log), gen_loss=np.array(losses), wall_s=np.array(wall),
                        params_json=np.array(json.dumps(params)), seed=np.array(seed),
                        probe_x=npy(xp[:, 0, :]), probe_t=npy(tp[0, :, 0]), probe_u=npy(up))
    print('wrote', path, 'steps', len(log), 'wall %.1fs' % wall, 'final rel-L2', log[-1])


def sphere_groups(case, domain_name, d, N_r, N_b, N_t, seed, funcs_module='configs.Ex4_1_funcs', radius=1.0, net=None, general_ac=False):
    """Time-varying ball domains (src/dataset.py:48-229): the sampled groups themselves, then one generator
    sub-iteration, one more, and one discriminator sub-iteration with the reference's own modules over EXACTLY the
    (interior, v, boundary) triples its training loop visits -- `for (datau, datav, bdata) in points`
    (src/training.py:128,153) iterates Comb_loader.__getitem__(k) for k = 0, 1, ... until the first IndexError
    (src/dataset.py:314-322), i.e. the natural pairing (k, k) up to the shorter of the two lists, INCLUDING the
    single-slice group 0 at T0 (where NeuralODE.forward returns [N,1], src/model.py:89-91, and loss.I / init / bdry
    broadcast to [N,N] pairwise terms, src/loss.py:65,70,79,84) and the T0 boundary group.  zero_grad() once per
    sub-iteration, optimizer.step() after every group (src/training.py:127-138,152-162); GPU loader semantics (fresh
    copies per pass)."""
    training, dataset, lossmod, F = load_reference(funcs_module, d)
    params = make_params(d, N_r, N_b, N_t, 'midpoint')
    params['domain'] = domain_name
    params['shape_param'] = radius
    if not funcs_module.endswith('Ex4_1_funcs'):
        params['funcs'] = funcs_module.split('.')[-1]      # (read back by the tests to pick the callables)
    if net is not None:
        params.update(net)
    if general_ac:
        # round 5: a general diffusion tensor a_ij(t, x) and a non-linear reaction c(u, t, x) (tests/golden/general_funcs.py) on a list
        # domain; b stays zero -- on the single-slice groups the reference sums the b-term with np.sum over a list of [N, N] tensors
        # (src/loss.py:69), which does not mean the same thing on every numpy
        import types
        sys.path.insert(0, HERE)
        import general_funcs as GF
        if general_ac == 'const':                  # (one constant matrix a, the linear reaction c = -0.7 u: the group runner's fused path)
            F = types.SimpleNamespace(func_a=GF.const_a, func_b=F.func_b, func_c=GF.lin_c, func_h=F.func_h, func_f=F.func_f,
                                      func_g=F.func_g, func_u_sol=F.func_u_sol)
            params['funcs'] = params.get('funcs', 'Ex4_1_funcs') + '+general_const'
        else:
            F = types.SimpleNamespace(func_a=GF.func_a, func_b=F.func_b, func_c=GF.func_c, func_h=F.func_h, func_f=F.func_f,
                                      func_g=F.func_g, func_u_sol=F.func_u_sol)
            params['funcs'] = params.get('funcs', 'Ex4_1_funcs') + '+general_ac'
    dev = torch.device('cpu')
    torch.manual_seed(seed)
    np.random.seed(seed)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, dev, './',
                                 func_u_sol=F.func_u_sol, p=2)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(seed)}
    domain = S.domain(S.setup['shape_param'], d, S.setup['T0'], S.setup['T'], N_t)
    points = dataset.Comb_loader(N_r, N_b, domain, dev)
    out['times'] = npy(domain.times)
    out['V'] = np.array(float(domain.V()))
    out['n_interior'], out['n_boundary'] = np.array(len(points.interioru)), np.array(len(points.boundary))
    for k, g in enumerate(points.interioru):
        out['interior/%d' % k] = npy(g)
        out['w/%d' % k] = npy(domain.func_w(g))
    for k, g in enumerate(points.boundary):
        out['boundary/%d' % k] = npy(g)
    out['L2_start'] = np.array(training.L_norm(points.interioru, S.u_net, S.p, S.func_u_sol, domain.V(), N_r).item())
    out['rel_start'] = np.array(training.rel_err(points.interioru, S.u_net, S.func_u_sol, S.p, domain.V(), N_r).item())
    # the triples the reference's loop visits, through the loader's own iteration protocol
    n_visited = sum(1 for _ in points)
    pairs = [(k, k) for k in range(n_visited)]
    assert n_visited == min(len(points.interioru), len(points.boundary))
    out['pairs'] = np.array(pairs)
    fresh = lambda t: t.detach().clone().requires_grad_(True)  # noqa: E731
    step = 0
    for which, opt, net in (('u', S.optimizer_u, S.u_net), ('u', S.optimizer_u, S.u_net), ('v', S.optimizer_v, S.v_net)):
        opt.zero_grad()
        for (ki, kb) in pairs:
            datau, datav, bdata = fresh(points.interioru[ki]), fresh(points.interiorv[ki]), fresh(points.boundary[kb])
            pv, pu = S.v_net(datav), S.u_net(datau)
            h, f, g, a, b, c = training.func_eval(datau.clone().detach(), bdata.clone().detach(), S.setup, pu,
                                                  F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g)
            Lo = lossmod.loss(S.config['alpha'], a, b, c, h, f, g, S.setup, domain, dev)
            tag = 'step%d' % step
            if which == 'u':
                with torch.no_grad():
                    out[tag + '/init'] = np.array(Lo.init(pu).item())
                    out[tag + '/bdry'] = np.array(Lo.bdry(S.u_net, bdata).item())
            L = Lo.u(pu, pv, S.u_net, datau, datav, bdata) if which == 'u' else Lo.v(pu, pv, datau, datav)
            L.backward(retain_graph=True)
            out[tag + '/which'] = np.array(which)
            out[tag + '/loss'] = np.array(L.item())
            out[tag + '/u'] = npy(pu.reshape(pu.shape[0], -1))          # ([N,1] on the single-slice T0 group, else [N,L,1])
            out[tag + '/v'] = npy(pv.squeeze(2))
            # a parameter whose .grad is still None (the field's, while no group of this sub-iteration has integrated the
            # ODE: zero_grad() sets gradients to None on torch >= 2.0) is SKIPPED by Adam -- recorded as a mask
            out[tag + '/grad'] = np.concatenate([(npy(p.grad) if p.grad is not None else np.zeros(tuple(p.shape))).reshape(-1)
                                                 for p in net.parameters()])
            out[tag + '/has_grad'] = np.concatenate([np.full(p.numel(), p.grad is not None) for p in net.parameters()])
            opt.step()
            out[tag + '/after'] = np.concatenate([npy(p).reshape(-1) for p in net.parameters()])
            out[tag + '/adam_steps'] = np.array([int(opt.state[p]['step']) if p in opt.state and 'step' in opt.state[p] else 0
                                                 for p in net.parameters()])
            step += 1
    out['n_steps'] = np.array(step)
    path = os.path.join(HERE, case + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024), 'groups', [tuple(g.shape[:2]) for g in points.interioru],
          'boundary', [g.shape[0] for g in points.boundary], 'pairs', pairs)


def sphere_sampling(case, domain_name, d, N_r, N_b, N_t, seed, radius):
    """The groups a ball domain samples at a radius that is NOT a power of two (round 4): every other fixture uses
    shape_param = 1.0, where `r * (float32 expression)` and `r * (float32 expression).double()` coincide -- the hourglass
    bound of src/dataset.py:89-90 is the second form (`.double()` binds before `*`)."""
    training, dataset, lossmod, F = load_reference()
    params = make_params(d, N_r, N_b, N_t, 'midpoint')
    params['domain'] = domain_name
    params['shape_param'] = radius
    torch.manual_seed(seed)
    np.random.seed(seed)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cpu'), './',
                                 func_u_sol=F.func_u_sol, p=2)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(seed)}
    domain = S.domain(S.setup['shape_param'], d, S.setup['T0'], S.setup['T'], N_t)
    points = dataset.Comb_loader(N_r, N_b, domain, torch.device('cpu'))
    out['times'] = npy(domain.times)
    out['V'] = np.array(float(domain.V()))
    out['n_interior'], out['n_boundary'] = np.array(len(points.interioru)), np.array(len(points.boundary))
    for k, g in enumerate(points.interioru):
        out['interior/%d' % k] = npy(g)
        out['w/%d' % k] = npy(domain.func_w(g))
    for k, g in enumerate(points.boundary):
        out['boundary/%d' % k] = npy(g)
    path = os.path.join(HERE, case + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024), 'groups', len(points.interioru), len(points.boundary))


def sphere_trajectory(case, domain_name, d, N_r, N_b, N_t, seed, outer_iters, funcs_module='configs.Ex4_3_funcs', alpha=100000000,
                      solver_name='midpoint', net=None, general_ac=False, hook_draws=False):
    """The reference's own train() on a time-varying ball domain (natural group loop incl. the single-slice groups),
    GPU loader semantics.  Its on-sample diagnostic is unusable on list domains (utils/auxillary_funcs.py:19 broadcasts
    [N,1] - [N] to [N,N] on single-slice groups), so the `stop` hook -- called once per generator sub-iteration,
    src/training.py:142 -- evaluates u_net on a FIXED multi-slice probe group that starts at T0 and stays strictly
    inside the domain, against func_u_sol: relative L2 error over the probe points."""
    training, dataset, lossmod, F = load_reference(funcs_module, d)
    params = make_params(d, N_r, N_b, N_t, solver_name, iterations=outer_iters, alpha=alpha)
    params['domain'] = domain_name
    params['shape_param'] = 1.0
    if net is not None:
        params.update(net)
    if not funcs_module.endswith('Ex4_1_funcs'):
        params['funcs'] = funcs_module.split('.')[-1]
    if general_ac:                                 # (as sphere_groups: general a_ij, c(u, t, x); b stays zero)
        import types
        sys.path.insert(0, HERE)
        import general_funcs as GF
        F = types.SimpleNamespace(func_a=GF.func_a, func_b=F.func_b, func_c=GF.func_c, func_h=F.func_h, func_f=F.func_f,
                                  func_g=F.func_g, func_u_sol=F.func_u_sol)
        params['funcs'] = params.get('funcs', 'Ex4_1_funcs') + '+general_ac'
    orig = dataset.Comb_loader.__getitem__

    def getitem(self, idx):
        r = orig(self, idx)
        return tuple(t.clone() for t in r)
    dataset.Comb_loader.__getitem__ = getitem
    # probe group: 96 points of radius < 0.45, 9 times in [0, 0.5]  (cone radius at t = 0.5 is 0.5; hourglass too)
    g = torch.Generator().manual_seed(4242)
    xp = torch.randn(96, d, generator=g, dtype=torch.float64)
    xp = xp / xp.norm(dim=1, keepdim=True) * (0.45 * torch.rand(96, 1, generator=g, dtype=torch.float64) ** (1.0 / d))
    tp = torch.linspace(0, 0.5, 9, dtype=torch.float64)
    probe = torch.cat((tp.view(1, -1, 1).expand(96, -1, 1), xp.view(96, 1, d).expand(-1, 9, -1)), 2).contiguous()
    sol = F.func_u_sol(probe)
    log, losses = [], []

    def hook(self, pts, domain):
        with torch.no_grad():
            up = self.u_net(probe).squeeze(2)
            log.append(float(torch.sqrt(torch.mean((up - sol) ** 2) / torch.mean(sol ** 2))))
        losses.append(self.av_l)
        if hook_draws:
            torch.rand(3)
            np.random.rand(2)
        return False
    torch.manual_seed(seed)
    np.random.seed(seed)
    cwd = os.getcwd()
    scratch = tempfile.mkdtemp(prefix='ref_traj_')
    os.chdir(scratch)
    try:
        S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g,
                                     torch.device('cpu'), './', func_u_sol=F.func_u_sol, p=2, stop=hook)
        t0 = time.time()
        S.train(report=False)
        wall = time.time() - t0
        with torch.no_grad():
            up = S.u_net(probe).squeeze(2)
    finally:
        os.chdir(cwd)
        dataset.Comb_loader.__getitem__ = orig
    path = os.path.join(HERE, case + '.npz')
    np.savez_compressed(path, rel_l2=np.array(log), gen_loss=np.array(losses), wall_s=np.array(wall),
                        params_json=np.array(json.dumps(params)), seed=np.array(seed), probe=npy(probe), probe_sol=npy(sol),
                        probe_u=npy(up))
    print('wrote', path, 'steps', len(log), 'wall %.1fs' % wall, 'rel-L2 first/last', log[0], log[-1])


def bound_pad_vectors():
    """NeuralODE.forward on paths that start neither at T0 nor on the boundary (src/model.py:92-106 with
    Hypercube.bound_pad / fillt): pins the evaluation path (proj plots with a fixed time use it)."""
    training, dataset, lossmod, F = load_reference()
    d = 3
    params = make_params(d, 8, 12, 6, 'midpoint')
    torch.manual_seed(7)
    np.random.seed(7)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cpu'), './',
                                 func_u_sol=F.func_u_sol, p=2)
    g = torch.Generator().manual_seed(5)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(7)}
    k = 0
    for times in ([0.3, 0.5, 0.9], [0.25, 0.26, 0.8, 1.0], [1.0, 1.0, 1.0]):
        x = torch.rand(5, 1, d, generator=g) - 0.5
        X = torch.cat((torch.tensor(times).view(1, -1, 1).repeat(5, 1, 1), x.repeat(1, len(times), 1)), 2)
        with torch.no_grad():
            u = S.u_net(X)
        out['%d/X' % k], out['%d/u' % k] = npy(X), npy(u)
        k += 1
    out['n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'ref_boundpad.npz'), **out)
    print('wrote ref_boundpad.npz', k, 'cases', [tuple(out['%d/u' % i].shape) for i in range(k)])


def bound_pad_hourglass_vectors():
    """The same evaluation path on the hourglass domain (NSphere_THourglass.bound_pad, src/dataset.py:127-152: per-path
    padded grids bucketed by length).  The reference only survives some inputs (e.g. it raises when no gap of the
    prepended time vector exceeds (T - T0) / N_t); the cases below are ones it evaluates."""
    training, dataset, lossmod, F = load_reference()
    d = 3
    params = make_params(d, 8, 12, 6, 'midpoint')
    params.update({'domain': 'NSphere_THourglass', 'shape_param': 1.0})
    torch.manual_seed(7)
    np.random.seed(7)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cpu'), './',
                                 func_u_sol=F.func_u_sol, p=2)
    g = torch.Generator().manual_seed(5)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(7)}
    k = 0
    for times, scale in (([0.6, 0.8, 0.9], 0.4), ([0.7, 0.7, 0.7], 0.4), ([0.3, 0.6, 0.9], 0.4), ([0.7], 0.4),
                         ([0.75, 0.95], -0.7), ([0.72, 0.8, 1.0], -0.7), ([0.9, 1.0], -0.85)):
        if scale > 0:
            x = (torch.rand(7, 1, d, generator=g) - 0.5) * scale
        else:       # radii spread up to |scale|: paths that left the domain and re-entered at different times |x| / r
            x = torch.randn(7, 1, d, generator=g)
            x = x / torch.sqrt(torch.sum(x ** 2, 2, keepdim=True)) * torch.linspace(0.2, -scale, 7).view(7, 1, 1)
        X = torch.cat((torch.tensor(times).view(1, -1, 1).repeat(7, 1, 1), x.repeat(1, len(times), 1)), 2)
        try:
            with torch.no_grad():
                u = S.u_net(X)
        except (IndexError, RuntimeError) as exc:       # the reference does not survive every input on this branch
            print('   reference fails on', times, scale, '->', repr(exc)[:80])
            continue
        out['%d/X' % k], out['%d/u' % k] = npy(X), npy(u)
        k += 1
    out['n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'ref_boundpad_hourglass.npz'), **out)
    print('wrote ref_boundpad_hourglass.npz', k, 'cases', [tuple(out['%d/u' % i].shape) for i in range(k)])


def bound_pad_cone_vectors():
    """round 5: the evaluation path on the cone (NSphere_TCone.bound_pad, src/dataset.py:220-223), on the inputs the reference survives"""
    training, dataset, lossmod, F = load_reference()
    d = 3
    params = make_params(d, 8, 12, 6, 'midpoint')
    params.update({'domain': 'NSphere_TCone', 'shape_param': 1.0})
    torch.manual_seed(7)
    np.random.seed(7)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cpu'), './',
                                 func_u_sol=F.func_u_sol, p=2)
    g = torch.Generator().manual_seed(6)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(7)}
    k = 0
    for times, scale in (([0.3, 0.5, 0.9], 0.2), ([0.25, 0.26, 0.8, 1.0], 0.1), ([0.6, 0.6, 0.6], 0.3), ([0.4], 0.3),
                         ([0.1, 0.45], -0.5), ([0.2, 0.3, 0.35], -0.6), ([0.05, 0.1], -0.9)):
        if scale > 0:
            x = (torch.rand(7, 1, d, generator=g) - 0.5) * scale
        else:       # radii spread up to |scale|
            x = torch.randn(7, 1, d, generator=g)
            x = x / torch.sqrt(torch.sum(x ** 2, 2, keepdim=True)) * torch.linspace(0.1, -scale, 7).view(7, 1, 1)
        X = torch.cat((torch.tensor(times).view(1, -1, 1).repeat(7, 1, 1), x.repeat(1, len(times), 1)), 2)
        try:
            with torch.no_grad():
                u = S.u_net(X)
        except (IndexError, RuntimeError, ValueError) as exc:
            print('   reference fails on', times, scale, '->', repr(exc)[:80])
            continue
        out['%d/X' % k], out['%d/u' % k] = npy(X), npy(u)
        k += 1
    out['n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'ref_boundpad_cone.npz'), **out)
    print('wrote ref_boundpad_cone.npz', k, 'cases', [tuple(out['%d/u' % i].shape) for i in range(k)])


def stop_taken(case, d, N_r, N_b, N_t, seed, fire_at):
    """round 5: the reference's own train() with a `stop` hook that returns True at its `fire_at`-th call (src/training.py:142-146:
    save the generator's weights under <path>, print, exit()): the loss list it leaves, the weights it saved, the files that exist."""
    training, dataset, lossmod, F = load_reference()
    params = make_params(d, N_r, N_b, N_t, 'midpoint', iterations=6)
    orig = dataset.Comb_loader.__getitem__
    dataset.Comb_loader.__getitem__ = lambda self, idx: tuple(t.clone() for t in orig(self, idx))   # (.to(device) on a GPU copies)
    calls = []

    def hook(self, pts, domain):
        calls.append(1)
        return len(calls) == fire_at
    torch.manual_seed(seed)
    np.random.seed(seed)
    cwd = os.getcwd()
    scratch = tempfile.mkdtemp(prefix='ref_stop_')
    os.chdir(scratch)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(seed), 'fire_at': np.array(fire_at)}
    try:
        S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cpu'),
                                     scratch + os.sep + 'run_', func_u_sol=F.func_u_sol, p=2, stop=hook)
        left = False
        try:
            S.train(report=False)
        except SystemExit:
            left = True
        assert left and len(calls) == fire_at
        out['losses'] = np.array(json.load(open('losses_NODE_%d.json' % d)))
        out['n_times'] = np.array(len(json.load(open('Time_NODE_%d.json' % d))))
        out['n_L2'] = np.array(len(json.load(open('L2_NODE_%d.json' % d))))
        sd = torch.load(scratch + os.sep + 'run_best_model_weights_NODE.pth')
        out['saved_keys'] = np.array(list(sd.keys()))
        for k_, v_ in sd.items():
            out['saved/' + k_] = npy(v_)
        best = torch.load('best_model_weights_NODE.pth')
        for k_, v_ in best.items():
            out['best/' + k_] = npy(v_)
    finally:
        os.chdir(cwd)
        dataset.Comb_loader.__getitem__ = orig
    np.savez_compressed(os.path.join(HERE, case + '.npz'), **out)
    print('wrote', case, 'losses', out['losses'])


def proj_vectors():
    """round 5: the reference's own plotting helper (utils/auxillary_funcs.py:34-98) on a freshly initialised solver: the two arrays
    it saves (guess_cn.npy, error_cn.npy) for a (t, x_1) slice -- a path tensor over the plot's own time grid -- and for an
    (x_1, x_2) slice at the fixed time T, where every row starts off T0 (bound_pad / fillt inside u_net)."""
    training, dataset, lossmod, F = load_reference()
    aux = sys.modules['utils.auxillary_funcs']
    d = 3
    params = make_params(d, 8, 12, 6, 'midpoint')
    torch.manual_seed(9)
    np.random.seed(9)
    S = training.NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cpu'), './',
                                 func_u_sol=F.func_u_sol, p=2)
    out = {'params_json': np.array(json.dumps(params)), 'seed': np.array(9)}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix='proj_')
    os.chdir(tmp)
    try:
        for k, axes in enumerate(([0, 1], [1, 2], [0, 3])):
            aux.proj(S.u_net, S.setup, 7, torch.device('cpu'), axes=list(axes), resolution=12, colours=6, save=True, show=False,
                     func_u_sol=F.func_u_sol)
            out['%d/axes' % k] = np.array(axes)
            out['%d/guess' % k] = np.load('guess_cn.npy')
            out['%d/error' % k] = np.load('error_cn.npy')
            assert os.path.exists('plot_at_7_along_' + str(list(axes)) + '.png')
        out['n'] = np.array(3)
    finally:
        os.chdir(cwd)
    # the error norms the loop and the configs' stop rule use (utils/auxillary_funcs.py:7-30) at p = 1, 2, 3, on a cube sample
    # (one tensor) and on a cone sample (a list of groups, weighted by their share of N_r)
    for tag, shape in (('cube', dataset.Hypercube(params['shape_param'], d, 0, 1, params['N_t'])), ('cone', dataset.NSphere_TCone(1.0, d, 0, 1, params['N_t']))):
        torch.manual_seed(21)
        np.random.seed(21)
        pts = dataset.Comb_loader(40, 24, shape, torch.device('cpu'))
        for p_ in (1, 2, 3):
            with torch.no_grad():
                out['%s/L%d' % (tag, p_)] = npy(aux.L_norm(pts.interioru, S.u_net, p_, F.func_u_sol, shape.V(), 40))
                out['%s/rel%d' % (tag, p_)] = npy(aux.rel_err(pts.interioru, S.u_net, F.func_u_sol, p_, shape.V(), 40))
    np.savez_compressed(os.path.join(HERE, 'ref_proj.npz'), **out)
    print('wrote ref_proj.npz', [tuple(out['%d/guess' % i].shape) for i in range(3)])


def fillt_vectors():
    """src/dataset.py:13-32 on a few hand-picked time vectors (the helper has surprising edge behaviour that the
    product reproduces verbatim: it can drop a sample and return indices past the filled vector)."""
    training, dataset, lossmod, F = load_reference()
    probes = [[0.0, 0.05, 0.6, 1.0], [0.0, 0.3, 0.35, 0.9, 1.0], [0.2, 0.25, 0.8], [0.0, 1.0], [0.0, 0.1, 0.2, 0.3]]
    out = {}
    k = 0
    for t in probes:
        for ms in (5, 20):
            i, f = dataset.fillt(torch.tensor(t), 1.0, 0.0, ms)
            out['%d/t' % k], out['%d/ms' % k], out['%d/idx' % k], out['%d/filled' % k] = np.array(t, dtype=np.float32), np.array(ms), npy(i), npy(f)
            k += 1
    out['n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'ref_fillt.npz'), **out)
    print('wrote ref_fillt.npz', k, 'cases')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--traj', action='store_true', help='also produce the 800-step trajectory fixtures (slow)')
    ap.add_argument('--only-traj', action='store_true')
    ap.add_argument('--round2', action='store_true', help='only the fixtures added in round 2 (BASELINE configs[2], configs[4])')
    ap.add_argument('--round3', action='store_true', help='only the fixtures added / regenerated in round 3 (natural group pairing on the '
                    'ball domains, d = 100, the cone trajectory)')
    ap.add_argument('--traj-d20', action='store_true', help='only the d = 20 cube trajectory fixture (round 3)')
    ap.add_argument('--round4', action='store_true', help='only the fixture added in round 4: one outer iteration of the reference at the '
                    'BENCHMARKED size (BASELINE configs[1]: d = 20, N_r = N_b = 4096, N_t = 32), slim record (~4 min, 1 GB)')
    ap.add_argument('--traj-hourglass', action='store_true', help='only the hourglass trajectory fixture (round 3, second ball domain)')
    ap.add_argument('--shapes', action='store_true', help='round 4: one outer iteration of the reference at three other network shapes')
    ap.add_argument('--general', action='store_true', help='round 4: one outer iteration of the reference with general a_ij, b_i, c(u,t,x)')
    ap.add_argument('--loops', action='store_true', help='round 5: the reference\'s train() with other sub-iteration counts (n1, n2) = (3, 2), (1, 3)')
    ap.add_argument('--alpha1', action='store_true', help='round 5: penalty weight alpha = 1 -- the interior loss term is not hidden behind alpha x penalties in the generator sub-steps')
    ap.add_argument('--intervals', action='store_true', help='round 5: one outer iteration of the reference on a time interval and a cube other than [0, 1] x [-1, 1]^d')
    ap.add_argument('--proj', action='store_true', help='round 5: the arrays the reference\'s proj() saves on a freshly initialised solver')
    ap.add_argument('--generic', action='store_true', help='round 5: one outer iteration of the reference at network widths beyond the '
                    'MFMA kernel instantiations (the generic path of csrc/xw_generic.hip)')
    ap.add_argument('--traj-cfg5', action='store_true', help='round 4: 8 outer iterations of the reference\'s own train() on both ball '
                    'domains at BASELINE config 5 size (Ex4_3, d = 10, N_r = N_b = 8192, N_t = 20)')
    ap.add_argument('--traj-ball-solvers', action='store_true', help='round 4: 40 outer iterations of the reference\'s own train() on the ball '
                    'domains with the other two fixed-grid schemes (euler on the hourglass, rk4 on the cone; d = 3)')
    ap.add_argument('--traj-headline-solvers', action='store_true', help='round 4: 30 outer iterations of the reference at the benchmarked '
                    'size with solver euler and rk4 (~5 + ~12 min)')
    ap.add_argument('--traj-headline', action='store_true', help='round 4: 250 outer iterations of the reference\'s own train() at the '
                    'BENCHMARKED size (d = 20, N_r = N_b = 4096, N_t = 32), rel-L2 at every generator sub-step (~30 min, 1 GB)')
    args = ap.parse_args()
    torch.set_num_threads(4)
    if args.general:
        one_iteration('ref_general_d4_midpoint', 4, 83, 45, 9, 5, 'midpoint', True, general=True, alpha=1000.0)
        sys.exit(0)
    if args.alpha1:
        # every other fixture runs at alpha = 1e3 .. 1e8, where the generator's gradient is alpha x (penalty gradients) + a part in 1e3 .. 1e8
        # of d log(I^2)/d theta: with alpha = 1 that part IS the gradient
        one_iteration('ref_alpha1_d4_midpoint', 4, 48, 28, 6, 28, 'midpoint', True, alpha=1.0)
        one_iteration('ref_alpha1_d3_rk4', 3, 40, 24, 5, 29, 'rk4', True, alpha=1.0)
        one_iteration('ref_alpha1_general_d4_euler', 4, 48, 28, 6, 30, 'euler', True, general=True, alpha=1.0)
        # the smallest legal shapes: d = 2 (Ex4_1 needs x_1, x_2), two sample times (one step), seven paths; and N_t = 3 with rk4
        one_iteration('ref_min_d2_nt2_midpoint', 2, 7, 5, 2, 34, 'midpoint', True, alpha=1.0)
        one_iteration('ref_min_d2_nt3_rk4', 2, 17, 33, 3, 35, 'rk4', True, alpha=100.0)
        # general coefficients through the reference's own train(): cube (a, b, c general; 20 outer iterations, alpha = 1e3), cone and hourglass
        # (a, c general; 8 outer iterations, alpha = 1e2)
        trajectory('ref_traj_general_d3_seed36', 3, 64, 40, 8, 36, 20, True, net=dict(alpha=1000.0), general=True)
        sphere_trajectory('ref_traj_cone_general_d3_seed37', 'NSphere_TCone', 3, 128, 64, 8, 37, 8, alpha=100.0, general_ac=True)
        sphere_trajectory('ref_traj_hourglass_general_d3_seed38', 'NSphere_THourglass', 3, 128, 64, 8, 38, 8, alpha=100.0, general_ac=True)
        # a stop hook that draws from torch's and numpy's global generators at every call: every later sample of the run moves
        trajectory('ref_traj_hook_draws_d3_seed39', 3, 64, 40, 8, 39, 12, True, hook_draws=True)
        sphere_trajectory('ref_traj_cone_hook_draws_d3_seed40', 'NSphere_TCone', 3, 128, 64, 8, 40, 8, alpha=10000.0, hook_draws=True)
        # the table forms of a: one constant matrix / a diagonal a(x), each with the linear reaction c = -0.7 u (fused path)
        one_iteration('ref_const_a_d4_midpoint', 4, 48, 28, 6, 41, 'midpoint', True, general='const', alpha=10.0)
        one_iteration('ref_diag_a_d5_rk4', 5, 40, 24, 5, 42, 'rk4', True, general='diag', alpha=10.0)
        sphere_groups('ref_hourglass_const_a_groups', 'NSphere_THourglass', 3, 64, 40, 8, 43, 'configs.Ex4_3_funcs', net=dict(alpha=10.0), general_ac='const')
        sphere_groups('ref_cone_alpha1_groups', 'NSphere_TCone', 3, 64, 40, 8, 31, 'configs.Ex4_3_funcs', net=dict(alpha=1.0))
        sphere_groups('ref_hourglass_alpha1_groups', 'NSphere_THourglass', 3, 64, 40, 8, 32, 'configs.Ex4_3_funcs', net=dict(alpha=1.0))
        sphere_groups('ref_hourglass_alpha1_general_groups', 'NSphere_THourglass', 3, 64, 40, 8, 33, 'configs.Ex4_3_funcs', net=dict(alpha=1.0), general_ac=True)
        sys.exit(0)
    if args.intervals:
        # T0 = 0.25, T = 1.5 and the asymmetric cube [-0.5, 1.5]^d (midpoint); T0 = -1, T = 0 on [0, 2]^d (rk4)
        one_iteration('ref_interval_d4_midpoint', 4, 48, 28, 6, 19, 'midpoint', True, shape_param=[-0.5, 1.5], net=dict(T0=0.25, T=1.5))
        one_iteration('ref_interval_d3_rk4', 3, 40, 24, 5, 20, 'rk4', True, shape_param=[0.0, 2.0], net=dict(T0=-1.0, T=0.0))
        # and 15 outer iterations of the reference's own train() there (the compact cube sample, its weight kernel, the refill graph)
        trajectory('ref_traj_interval_d3_seed21', 3, 64, 40, 8, 21, 15, True, net=dict(T0=0.25, T=1.5, shape_param=[-0.5, 1.5]))
        # the ball domains at radius 0.7 (10 outer iterations each)
        # one sample at radius 0.7 whose last visited boundary group is ONE path on one time slice (seed 101), group by group
        sphere_groups('ref_cone_r07_groups', 'NSphere_TCone', 3, 128, 64, 8, 101, 'configs.Ex4_3_funcs', radius=0.7)
        # (the ball domains on another time interval are not a configuration the reference survives: T = 1.4 asks its cone sampler for a
        #  negative number of points, src/dataset.py:180,203-214, and T0 = 0.1 sends its single-slice groups into an IndexError, src/model.py:106)
        # general a_ij, c(u, t, x) on both ball domains (one sample each, group by group)
        sphere_groups('ref_cone_general_groups', 'NSphere_TCone', 3, 64, 40, 8, 26, 'configs.Ex4_3_funcs', general_ac=True)
        sphere_groups('ref_hourglass_general_groups', 'NSphere_THourglass', 3, 64, 40, 8, 27, 'configs.Ex4_3_funcs', general_ac=True)
        sphere_trajectory('ref_traj_cone_r07_d3_seed22', 'NSphere_TCone', 3, 128, 64, 8, 22, 10, alpha=10000.0, net=dict(shape_param=0.7))
        sphere_trajectory('ref_traj_hourglass_r07_d3_seed23', 'NSphere_THourglass', 3, 128, 64, 8, 23, 10, alpha=10000.0, net=dict(shape_param=0.7))
        sys.exit(0)
    if args.loops:
        # other sub-iteration counts than the YAML's n1 = 2, n2 = 1 (src/training.py:125,151): several discriminator sub-steps per outer
        # iteration move phi between them, several generator sub-steps see the same phi
        trajectory('ref_traj_n1_3_n2_2_d3_seed16', 3, 64, 40, 8, 16, 20, True, net=dict(n1=3, n2=2))
        trajectory('ref_traj_n1_1_n2_3_d3_seed17', 3, 64, 40, 8, 17, 30, True, net=dict(n1=1, n2=3))
        sphere_trajectory('ref_traj_cone_n1_3_n2_2_d3_seed18', 'NSphere_TCone', 3, 128, 64, 8, 18, 12, alpha=10000.0, net=dict(n1=3, n2=2))
        sys.exit(0)
    if args.proj:
        proj_vectors()
        bound_pad_cone_vectors()
        stop_taken('ref_stop_taken_d3_seed15', 3, 64, 40, 8, 15, 5)
        sys.exit(0)
    if args.generic:
        # wider than the stepper's (32, 12) and the test network's 64: what csrc/xw_generic.hip serves (up to (64, 16) / 128)
        one_iteration('ref_generic_d5_midpoint', 5, 48, 28, 6, 11, 'midpoint', True,
                      net=dict(u_hidden_dim=48, u_hidden_hidden_dim=16, u_layers=4, v_hidden_dim=100, v_layers=3))
        one_iteration('ref_generic_d3_rk4', 3, 36, 20, 5, 12, 'rk4', True,
                      net=dict(u_hidden_dim=64, u_hidden_hidden_dim=16, u_layers=2, v_hidden_dim=128, v_layers=2))
        one_iteration('ref_generic_mixed_d4_euler', 4, 40, 24, 5, 13, 'euler', True,
                      net=dict(u_hidden_dim=20, u_hidden_hidden_dim=10, u_layers=8, v_hidden_dim=70, v_layers=9))
        # and 25 outer iterations of the reference's own train() at (48, 16) / 100
        trajectory('ref_traj_generic_d3_seed14', 3, 64, 40, 8, 14, 25, True,
                   net=dict(u_hidden_dim=48, u_hidden_hidden_dim=16, u_layers=4, v_hidden_dim=100, v_layers=3))
        sys.exit(0)
    if args.shapes:
        # the widest / deepest networks the engine compiles (stepper container (32, 12), depth 10; test network width 64), a
        # narrow odd-sized pair that runs zero-padded inside the (20, 10) / 50 containers, and a field without hidden layer
        one_iteration('ref_wide_d6_midpoint', 6, 70, 40, 7, 8, 'midpoint', True,
                      net=dict(u_hidden_dim=32, u_hidden_hidden_dim=12, u_layers=10, v_hidden_dim=64, v_layers=4))
        one_iteration('ref_narrow_d3_euler', 3, 50, 30, 6, 9, 'euler', True,
                      net=dict(u_hidden_dim=7, u_hidden_hidden_dim=3, u_layers=2, v_hidden_dim=11, v_layers=2))
        one_iteration('ref_m1_d4_rk4', 4, 40, 24, 5, 10, 'rk4', True,
                      net=dict(u_hidden_dim=16, u_hidden_hidden_dim=8, u_layers=1, v_hidden_dim=57, v_layers=12))
        sys.exit(0)
    if args.traj_cfg5:
        # BASELINE configs[4] AT ITS STATED SIZE (Ex4_3, d = 10, N_r = N_b = 8192, N_t = 20, alpha = 1e4 as tools/train_cfg5.py runs it):
        # 8 outer iterations of the reference's own train() per ball domain
        sphere_trajectory('ref_traj_cone_ex43_d10_full_seed2', 'NSphere_TCone', 10, 8192, 8192, 20, 2, 8, alpha=10000.0)
        sphere_trajectory('ref_traj_hourglass_ex43_d10_full_seed3', 'NSphere_THourglass', 10, 8192, 8192, 20, 3, 8, alpha=10000.0)
        sys.exit(0)
    if args.traj_ball_solvers:
        sphere_trajectory('ref_traj_hourglass_ex43_d3_euler_seed7', 'NSphere_THourglass', 3, 256, 128, 10, 7, 40, solver_name='euler')
        sphere_trajectory('ref_traj_cone_ex43_d3_rk4_seed8', 'NSphere_TCone', 3, 256, 128, 10, 8, 40, solver_name='rk4')
        sys.exit(0)
    if args.traj_headline_solvers:
        # the other two fixed-grid schemes at the benchmarked size: 30 outer iterations of the reference's own train() each
        trajectory('ref_traj_d20_headline_euler_seed5', 20, 4096, 4096, 32, 5, 30, True, solver_name='euler')
        trajectory('ref_traj_d20_headline_rk4_seed6', 20, 4096, 4096, 32, 6, 30, True, solver_name='rk4')
        sys.exit(0)
    if args.traj_headline:
        trajectory('ref_traj_d20_headline_seed4', 20, 4096, 4096, 32, 4, 250, True)
        sys.exit(0)
    if args.traj_d20:
        # trained-error parity at the headline dimension (BASELINE configs[1] family: d = 20; N small enough for the reference)
        trajectory('ref_traj_d20_seed2_gpusem', 20, 128, 96, 12, 2, 150, True)
        sys.exit(0)
    if args.round4:
        t0 = time.time()
        one_iteration('ref_d20_headline', 20, 4096, 4096, 32, 11, 'midpoint', False, slim=64)
        print('reference time %.0f s' % (time.time() - t0))
        sphere_sampling('ref_hourglass_r07_sampling', 'NSphere_THourglass', 3, 400, 64, 64, 5, 0.7)
        sphere_sampling('ref_cone_r07_sampling', 'NSphere_TCone', 3, 400, 64, 64, 5, 0.7)
        sys.exit(0)
    if args.traj_hourglass:
        sphere_trajectory('ref_traj_hourglass_ex43_d3_seed1', 'NSphere_THourglass', 3, 256, 128, 10, 1, 60)
        sys.exit(0)
    if args.round3:
        # BASELINE configs[3] shape family: d = 100, N_t = 32 (N small enough for the reference's a[d,d,N,L] table: 20 MB)
        one_iteration('ref_d100_small_midpoint', 100, 16, 64, 32, 5, 'midpoint', False, shape_param=[-1.0, 1.0])
        # ball domains with the NATURAL (k, k) pairing of the reference's loop, single-slice T0 groups included
        sphere_groups('ref_cone_groups', 'NSphere_TCone', 3, 64, 40, 8, 1)
        sphere_groups('ref_hourglass_groups', 'NSphere_THourglass', 3, 64, 40, 8, 1)
        sphere_groups('ref_cone_ex43_d10_groups', 'NSphere_TCone', 10, 384, 120, 12, 2, 'configs.Ex4_3_funcs')
        sphere_groups('ref_hourglass_ex43_d10_groups', 'NSphere_THourglass', 10, 384, 120, 12, 2, 'configs.Ex4_3_funcs')
        # trained error on a ball domain through the reference's own train()
        sphere_trajectory('ref_traj_cone_ex43_d3_seed0', 'NSphere_TCone', 3, 256, 128, 10, 0, 100)
        sphere_trajectory('ref_traj_hourglass_ex43_d3_seed1', 'NSphere_THourglass', 3, 256, 128, 10, 1, 60)
        trajectory('ref_traj_d20_seed2_gpusem', 20, 128, 96, 12, 2, 150, True)
        sys.exit(0)
    if args.round2 or not args.only_traj:
        # BASELINE configs[2] shape family: d = 50, N_t = 64 (small N so that the reference runs in seconds)
        one_iteration('ref_d50_nt64_small_midpoint', 50, 32, 100, 64, 4, 'midpoint', False)
        # BASELINE configs[4]: the time-varying ball domains with the Ex4_3 functions at d = 10
        sphere_groups('ref_cone_ex43_d10_groups', 'NSphere_TCone', 10, 384, 120, 12, 2, 'configs.Ex4_3_funcs')
        sphere_groups('ref_hourglass_ex43_d10_groups', 'NSphere_THourglass', 10, 384, 120, 12, 2, 'configs.Ex4_3_funcs')
        one_iteration('ref_d100_small_midpoint', 100, 16, 64, 32, 5, 'midpoint', False, shape_param=[-1.0, 1.0])
    if not args.only_traj and not args.round2:
        fillt_vectors()
        bound_pad_vectors()
        bound_pad_hourglass_vectors()
        sphere_groups('ref_cone_groups', 'NSphere_TCone', 3, 64, 40, 8, 1)
        sphere_groups('ref_hourglass_groups', 'NSphere_THourglass', 3, 64, 40, 8, 1)
        one_iteration('ref_tiny_midpoint', 3, 8, 12, 6, 7, 'midpoint', True)
        one_iteration('ref_tiny_euler', 3, 8, 12, 6, 7, 'euler', True)
        one_iteration('ref_tiny_rk4', 3, 8, 12, 6, 7, 'rk4', True)
        one_iteration('ref_plumb_midpoint', 5, 256, 64, 16, 0, 'midpoint', False)
        one_iteration('ref_d20_small_midpoint', 20, 96, 80, 12, 3, 'midpoint', False)
    if args.traj or args.only_traj:
        trajectory('ref_traj_plumb_seed0_gpusem', 5, 256, 64, 16, 0, 400, True)
        trajectory('ref_traj_plumb_seed0_cpusem', 5, 256, 64, 16, 0, 400, False)
        sphere_trajectory('ref_traj_cone_ex43_d3_seed0', 'NSphere_TCone', 3, 256, 128, 10, 0, 100)
