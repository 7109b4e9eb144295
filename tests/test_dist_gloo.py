"""world_size-2 `gloo` test (CPU) of the multi-GPU exchange: path sharding (dist.World.shard_group / bounds), the
32-byte partial-sum all-reduce and the packed-gradient all-reduce.  The per-shard arithmetic is done with the oracle
(the HIP kernels need a GPU); what is under test is that shard-local partials with GLOBAL 1/N factors, combined by
dist.World exactly the way engine.Engine combines them, reproduce the unsharded loss and gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import configs.Ex4_1_funcs as P

FUNCS = dict(h=P.func_h, f=P.func_f, g=P.func_g, a=P.func_a, b=P.func_b, c=P.func_c)
PARAMS = {'alpha': 1e8, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9, 'v_hidden_dim': 50,
          'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False, 'solver': 'midpoint',
          'dim': 4, 'N_t': 6, 'N_r': 37, 'N_b': 21, 'T0': 0, 'T': 1, 'shape_param': [-1, 1], 'iterations': 1,
          'domain': 'Hypercube'}


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _sharded_substep(which, O, world):
    """what one rank of the engine does in a sub-step, with oracle arithmetic on its shard"""
    from oracle import refspec as R
    X, XV, BX, n, nb = world.shard_group(O.X, O.XV, O.BX)
    theta = {k: v.clone().requires_grad_(which == 'u') for k, v in O.theta.items()}
    phi = {k: v.clone().requires_grad_(which == 'v') for k, v in O.phi.items()}
    o = R.forward_all(theta, phi, O.config, O.setup, O.cube, FUNCS, X, XV, BX, which == 'u')
    L = X.shape[1]
    I_part = R.weak_I(O.setup, o['V'], o['u'], o['v'], o['w'], o['du'], o['dphi'], o['h'], o['f'],
                      *R.tabulate(FUNCS, O.setup, X, BX, o['u'])[3:5], R.tabulate(FUNCS, O.setup, X, BX, o['u'])[5], n_glob=n)
    S_part = torch.sum(o['v'] ** 2)
    sse_i = torch.sum((o['u'][:, 0] - o['h']) ** 2)
    sse_b = torch.sum((o['u_b'] - o['g']) ** 2) if which == 'u' else torch.zeros((), dtype=torch.float64)
    scal = torch.stack([I_part, S_part, sse_i, sse_b]).detach().clone()
    world.all_reduce(scal)                                                   # exchange 1: 32 bytes
    I, S = scal[0], scal[1]
    alpha = O.config['alpha']
    if which == 'u':
        # d loss_u / d theta on this shard with the GLOBAL I:  (2/I) dI_part + alpha (d sse_i / n + d sse_b / (nb L)) + pollution
        surrogate = (2.0 / I) * I_part + alpha * (sse_i / n + sse_b / (nb * L))
        keys = list(theta)
        g = torch.autograd.grad(surrogate, [theta[k] for k in keys], allow_unused=True)
        flat = torch.cat([(o['pol_theta'][k] + (gi if gi is not None else 0)).reshape(-1) for k, gi in zip(keys, g)])
        loss = torch.log(I ** 2) - torch.log(o['V'] * S / (n * L)) + alpha * (scal[2] / n + scal[3] / (nb * L))
    else:
        surrogate = -(2.0 / I) * I_part + S_part / S
        keys = list(phi)
        g = torch.autograd.grad(surrogate, [phi[k] for k in keys], allow_unused=True)
        flat = torch.cat([(o['pol_phi'][k] + (gi if gi is not None else 0)).reshape(-1) for k, gi in zip(keys, g)])
        loss = -(torch.log(I ** 2) - torch.log(o['V'] * S / (n * L)))
    flat = flat.detach().clone()
    world.all_reduce(flat)                                                   # exchange 2: packed gradient
    return loss.item(), flat, keys


def _worker(rank, size, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import dist as xdist
    world, _ = xdist.init_from_env('gloo')
    assert world.size == size and world.rank == rank
    torch.manual_seed(3)
    O = R.Solver(PARAMS, FUNCS, u_sol=P.func_u_sol, p=2)
    O.new_sample()                                       # same seed on every rank -> same global sample
    res = {}
    for which in ('u', 'v'):
        loss, flat, keys = _sharded_substep(which, O, world)
        ref = (R.generator_grad if which == 'u' else R.discriminator_grad)(O.theta, O.phi, O.config, O.setup, O.cube, FUNCS, O.X, O.XV, O.BX)
        ref_flat = torch.cat([ref['grad'][k].reshape(-1) for k in keys])
        res[which] = (loss, ref['loss'].item(), float((flat - ref_flat).abs().max()), float(ref_flat.abs().max()))
    lo, hi = world.bounds(37)
    res['bounds'] = (lo, hi)
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('size', [2, 8])
def test_two_rank_gloo_exchange_reproduces_unsharded_step(tmp_path, size):
    """(also at 8 ranks, the node size the job is stated for: shards of 4-5 interior and 2-3 boundary paths)"""
    mp.spawn(_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    res = [torch.load(tmp_path / ('rank%d.pt' % r)) for r in range(size)]
    if size == 2:
        assert res[0]['bounds'] == (0, 19) and res[1]['bounds'] == (19, 37)
    else:
        assert [r['bounds'][1] - r['bounds'][0] for r in res] == [5, 5, 5, 5, 5, 4, 4, 4] and res[-1]['bounds'][1] == 37
    for r in res:
        for which in ('u', 'v'):
            loss, ref_loss, err, scale = r[which]
            np.testing.assert_allclose(loss, ref_loss, rtol=1e-6 if which == 'v' else 1e-9)
            assert err <= 2e-5 * scale, (which, err, scale)
    assert all(r['u'][0] == res[0]['u'][0] for r in res)  # every rank sees the same global loss


def test_bounds_cover_everything_once():
    from xnode_wan_pde_solver_amd.dist import World

    class W(World):
        def __init__(self, rank, size):
            self.rank, self.size, self.group = rank, size, None
    # the headline batch on a node: eight shards of 512 paths; an uneven batch: the first ranks take the extra paths
    assert [W(r, 8).bounds(4096) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    assert [b - a for a, b in (W(r, 8).bounds(4099) for r in range(8))] == [513, 513, 513, 512, 512, 512, 512, 512]
    # fewer paths than ranks: empty shares, never an error (list domains: groups of 1-5 paths)
    assert [W(r, 8).bounds(3) for r in range(8)] == [(0, 1), (1, 2), (2, 3)] + [(3, 3)] * 5
    for n in (1, 7, 8, 4096, 4097):
        for size in (1, 2, 3, 8):
            spans = [W(r, size).bounds(n) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _sampling_worker(rank, size, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from xnode_wan_pde_solver_amd import dist as xdist, sampling
    world, _ = xdist.init_from_env('gloo')
    torch.manual_seed(21)                                 # the shared seed of the job
    N_r, N_b, d = 1001, 403, 5
    out = []
    for it in range(2):                                   # two outer iterations: domain (time grid) + rank-local sample
        dom = sampling.Hypercube([-1, 1], d, 0, 1, 7)
        pts = sampling.RankCubeLoader(N_r, N_b, dom, torch.device('cpu'), world.rank, world.size)
        t, xu, xv, xb = pts.compact()
        out.append(dict(t=t.clone(), xu=xu.clone(), xv=xv.clone(), xb=xb.clone(), n=pts.n_local, nb=pts.nb_local))
    out.append(dict(after=torch.rand(3)))                 # the shared stream is in the same place on every rank
    torch.save(out, os.path.join(out_dir, 'samp%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('size', [2, 8])
def test_rank_local_sampling_draws_only_the_shard(tmp_path, size):
    """sampling.RankCubeLoader (solver.rank_local_sampling): every rank draws its share only; shares add up to the global
    counts, the time grids and the shared RNG stream stay common, the points are different draws on every rank, interior
    points are uniform in the cube and boundary points sit on faces in the global proportions (2 ranks and a node's 8)"""
    mp.spawn(_sampling_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    rs = [torch.load(tmp_path / ('samp%d.pt' % r)) for r in range(size)]
    r0 = rs[0]
    assert all(torch.equal(r0[2]['after'], r[2]['after']) for r in rs)
    for it in range(2):
        shares = [r[it] for r in rs]
        assert sum(a['n'] for a in shares) == 1001 and sum(a['nb'] for a in shares) == 403
        assert max(a['n'] for a in shares) - min(a['n'] for a in shares) <= 1
        faces = torch.zeros(5, dtype=torch.float64)
        for a in shares:
            assert torch.equal(a['t'], shares[0]['t'])
            assert a['xu'].shape == (a['n'], 5) and a['xb'].shape == (a['nb'], 5)
            assert not torch.equal(a['xu'], a['xv'])
            assert float(a['xu'].abs().max()) <= 1.0 and abs(float(a['xu'].mean())) < (0.1 if size == 2 else 0.2)
            on_face = (a['xb'].abs() == 1.0)
            assert bool(torch.all(on_face.sum(1) >= 1))
            faces += on_face.sum(0).double()
        per_axis = faces / 403
        assert float(per_axis.min()) > 0.08 and float(per_axis.max()) < 0.35      # ~ 1 / d each, over the whole job
        assert not torch.equal(shares[0]['xu'][:50], shares[1]['xu'][:50])      # different draws on different ranks
    assert not torch.equal(r0[0]['xu'], r0[1]['xu'])       # a new draw every iteration


def _fallback_worker(rank, size, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    import warnings
    from xnode_wan_pde_solver_amd import dist as xdist, _lib
    torch.distributed.init_process_group('gloo')
    calls = {'init': 0}

    class Lib:                                             # the C ABI as one rank without RCCL would see it
        def xw_comm_available(self):
            return -4 if rank == 1 else 0

        def xw_comm_unique_id(self, buf):
            return 0

        def xw_comm_init(self, *a):
            calls['init'] += 1                             # collective in the real library: entering it alone = a hang
            return 0

        def xw_comm_destroy(self, c):
            return 0
    real = _lib.lib
    _lib.lib = Lib()
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            world = xdist.World(native=True)
    finally:
        _lib.lib = real
    t = torch.full((3,), float(rank + 1), dtype=torch.float64)
    world.all_reduce(t)                                    # the torch.distributed path still works
    torch.save(dict(comm=world.comm is not None, capturable=world.capturable, init_calls=calls['init'], warned=len(w), sum=t.clone()),
               os.path.join(out_dir, 'fb%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_rank_without_rccl_sends_every_rank_to_the_fallback_before_the_collective_init(tmp_path):
    """ncclCommInitRank is collective: the ranks agree on `xw_comm_available` first; if one of them cannot bind RCCL NONE of
    them enters xw_comm_init (the others would block in its bootstrap), all fall back with a warning"""
    size = 2
    mp.spawn(_fallback_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    for r in range(size):
        res = torch.load(tmp_path / ('fb%d.pt' % r))
        assert res['comm'] is False and res['capturable'] is False and res['init_calls'] == 0 and res['warned'] >= 1
        assert torch.equal(res['sum'], torch.full((3,), 3.0, dtype=torch.float64))


def _pairwise_worker(rank, size, port, out_dir, golden_dir):
    """a single-slice T0 group of a ball domain, sharded: what engine.Engine does per rank (factorised pair sums with
    GLOBAL means, the two factors of the d(phi)/dt term reduced before they are multiplied) against the oracle's literal
    [N,N] broadcast on the whole group"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import json
    import configs.Ex4_3_funcs as F
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import dist as xdist
    world, _ = xdist.init_from_env('gloo')
    z = np.load(os.path.join(golden_dir, 'ref_cone_ex43_d10_groups.npz'))
    params = json.loads(str(z['params_json']))
    params.pop('funcs')
    funcs = dict(h=F.func_h, f=F.func_f, g=F.func_g, a=F.func_a, b=F.func_b, c=F.func_c)
    config, setup = R.split_params(params)
    torch.manual_seed(int(z['seed']))
    torch.Tensor(setup['N_t']).uniform_(setup['T0'], setup['T'])
    theta, phi = R.init_parameters(config, setup)
    domain = R.Ball(params['domain'], setup['shape_param'], setup['dim'], setup['T0'], setup['T'])
    X, BX = torch.from_numpy(z['interior/0']), torch.from_numpy(z['boundary/0'])
    assert X.shape[1] == 1 and BX.shape[1] == 1
    ref = R.group_forward(theta, phi, config, setup, domain, funcs, X, X, BX, True)
    Xs, XVs, BXs, n, nb = world.shard_group(X, X, BX)
    # shard-local quantities (oracle arithmetic on the shard, elementwise shapes)
    Xl = Xs.clone().requires_grad_(True)
    XVl = XVs.clone().requires_grad_(True)
    u = R.u_net(theta, config, Xl, funcs['h'](Xl[:, 0, :]))[:, 0]
    v = R.v_net(phi, config, XVl)[:, 0]
    w = domain.func_w(XVl)[:, 0]
    du = torch.autograd.grad(u.sum(), Xl)[0][:, 0, 1:]
    dphi = torch.autograd.grad((v * w).sum(), XVl)[0][:, 0, :]
    h, f = funcs['h'](Xs[:, 0, :]), funcs['f'](Xs)[:, 0]
    ub = R.u_net(theta, config, BXs, funcs['h'](BXs[:, 0, :]))[:, 0]
    g = funcs['g'](BXs)[:, 0]
    stats = torch.stack([h.sum(), (h ** 2).sum(), f.sum(), g.sum(), (g ** 2).sum()]).detach()
    world.all_reduce(stats)                                # Engine.load_group: the means are over the whole group
    hbar, fbar, gbar = stats[0] / n, stats[2] / n, stats[3] / nb
    var_h, var_g = stats[1] / n - hbar ** 2, stats[4] / nb - gbar ** 2
    V = domain.V()
    s31 = (dphi[:, 1:] * du).sum(1)                        # a = identity
    cu = -u                                                # Ex4_3: c = -u
    part = torch.stack([(V / n * (u * v - h * v) + (V / n) * n * (s31 + cu * u * v * w + fbar * v * w)).sum(), (v ** 2).sum(),
                        ((u - hbar) ** 2).sum(), ((ub - gbar) ** 2).sum(), torch.zeros(()), torch.zeros(()), torch.zeros(()),
                        u.sum(), dphi[:, 0].sum()]).detach().to(torch.float64)
    world.all_reduce(part)                                 # Engine._reduce_sums: scal[0:9]
    I = part[0] - (V / n) * part[7] * part[8]              # xw_pair_fold
    init = part[2] / n + var_h
    bdry = part[3] / nb + var_g
    res = dict(I=(float(I), float(ref['I'])), init=(float(init), float(ref['init'])), bdry=(float(bdry), float(ref['bdry'])),
               S=(float(part[1]), float((ref['v3'] ** 2).sum())))
    torch.save(res, os.path.join(out_dir, 'pw%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_pairwise_group_reproduces_the_broadcast_of_the_whole_group(tmp_path, golden_dir):
    size = 2
    mp.spawn(_pairwise_worker, args=(size, _free_port(), str(tmp_path), golden_dir), nprocs=size, join=True)
    for r in range(size):
        res = torch.load(tmp_path / ('pw%d.pt' % r))
        for k, (got, want) in res.items():
            np.testing.assert_allclose(got, want, rtol=1e-9, err_msg=k)


# ---- list domains under a world: groups with fewer paths than ranks (BASELINE configs[4] is stated on 4 GPUs) ------------------
GROUP_SIZES = [5, 3, 3, 4, 1, 2]          # interior paths kept per group of the fixture (a cone sample's late groups look like this)
BDRY_SIZES = [4, 3, 2, 1, 4, 3]           # boundary paths kept (one group of a single path)


def _tiny_groups(golden_dir):
    """the six (interior, boundary) groups of the reference's cone fixture, cut down to a handful of paths each"""
    import json
    from oracle import refspec as R
    z = np.load(os.path.join(golden_dir, 'ref_cone_groups.npz'))
    params = json.loads(str(z['params_json']))
    params.pop('funcs', None)
    config, setup = R.split_params(params)
    torch.manual_seed(int(z['seed']))
    torch.Tensor(setup['N_t']).uniform_(setup['T0'], setup['T'])
    theta, phi = R.init_parameters(config, setup)
    domain = R.Ball(params['domain'], setup['shape_param'], setup['dim'], setup['T0'], setup['T'])
    triples = []
    for k, (n, nb) in enumerate(zip(GROUP_SIZES, BDRY_SIZES)):
        X, BX = torch.from_numpy(z['interior/%d' % k])[:n].clone(), torch.from_numpy(z['boundary/%d' % k])[:nb].clone()
        assert X.shape[0] == n and BX.shape[0] == nb
        triples.append((X, X, BX))
    return config, setup, theta, phi, domain, triples


def _rank_substep(which, theta, phi, config, setup, domain, funcs, triple, world, carried, touched):
    """what ONE RANK does in one group's sub-step of a list domain -- engine.Engine / csrc/xw_substep.hip with oracle arithmetic on
    the rank's share: shares through dist.World.shard_group (possibly EMPTY), the time grid and the start kind from the WHOLE
    group's first path, partial sums with GLOBAL 1/N factors, zeros from a rank that holds nothing, the exchanges of dist.py, the
    pairwise single-slice form factorised, the field's parameters skipped by Adam on facts every rank shares"""
    from oracle import refspec as R
    F64 = torch.float64
    X, XV, BX = triple
    Xs, XVs, BXs, n, nb = world.shard_group(X, XV, BX)
    grid_first, bgrid_first = float(X[0, 0, 0]), float(BX[0, 0, 0])
    L, Lb, V, alpha, T0 = X.shape[1], BX.shape[1], domain.V(), config['alpha'], setup['T0']
    pair_i, pair_b = L == 1 and grid_first == T0, nb > 0 and Lb == 1 and bgrid_first == T0
    keys = list(theta if which == 'u' else phi)
    par = {k: v.clone().requires_grad_(True) for k, v in (theta if which == 'u' else phi).items()}
    th, ph = (par, phi) if which == 'u' else (theta, par)
    zero = torch.zeros((), dtype=F64)
    # Engine.load_group: the sample-only halves of the pair sums are means over the WHOLE group (one exchange per sample)
    hbar = fbar = gbar = var_h = var_g = 0.0
    if pair_i or pair_b:
        hs, fs = funcs['h'](Xs[:, 0, :]).double(), funcs['f'](Xs).double()
        gs = funcs['g'](BXs).double() if pair_b else torch.zeros(0, dtype=F64)
        stats = torch.stack([hs.sum(), (hs ** 2).sum(), fs.sum(), gs.sum(), (gs ** 2).sum()]).detach()
        world.all_reduce(stats)
        hbar, fbar, var_h = stats[0] / n, stats[2] / n, stats[1] / n - (stats[0] / n) ** 2
        if pair_b:
            gbar, var_g = stats[3] / nb, stats[4] / nb - (stats[3] / nb) ** 2
    A = B = None
    scal = torch.zeros(9, dtype=F64)
    part_I = part_S = sse = zero
    if Xs.shape[0] > 0:
        Xl, XVl = Xs.clone().requires_grad_(True), XVs.clone().requires_grad_(True)
        start = funcs['h'](Xl[:, 0, :]) if grid_first == T0 else funcs['g'](Xl[:, 0, :].unsqueeze(1)).reshape(-1)
        Xgrid = torch.cat([X[:1].detach(), Xl], 0) if L > 1 else Xl      # (path 0 of the WHOLE group carries the grid, src/model.py:92)
        u = R.u_net(th, config, Xgrid, torch.cat([start[:1].detach() * 0, start]) if L > 1 else start)
        u = u[1:] if L > 1 else u
        v = R.v_net(ph, config, XVl)
        w = domain.func_w(XVl)
        du = torch.autograd.grad(u.sum(), Xl, retain_graph=True)[0]
        dphi = torch.autograd.grad((v * w).sum(), XVl, retain_graph=True)[0]
        pol = torch.autograd.grad(u.sum() if which == 'u' else (v * w).sum(), [par[k] for k in keys], retain_graph=True, allow_unused=True)
        h, f, _, a, b, c = R.tabulate(funcs, setup, Xs, BXs if BXs.shape[0] else BX, u)
        wd = w.detach()
        if pair_i:
            s31 = sum(a[i, j][:, 0] * dphi[:, 0, i + 1] * du[:, 0, j + 1] for i in range(setup['dim']) for j in range(setup['dim']))
            u0, v0, w0 = u[:, 0], v[:, 0], wd[:, 0]
            part_I = ((V / n) * (u0 * v0 - h * v0) + (V / n) * n * (s31 + c[:, 0] * u0 * v0 * w0 + fbar * v0 * w0)).sum()
            sse = ((u0 - hbar) ** 2).sum()
            scal[7], scal[8] = u0.sum().detach(), dphi[:, 0, 0].sum()
        else:
            part_I = R.weak_I(setup, V, u, v, wd, du, dphi, h, f, a, b, c, n_glob=n)
            sse = ((u[:, 0] - h) ** 2).sum()
        part_S = (v ** 2).sum()
        scal[0], scal[1], scal[2] = part_I.detach(), part_S.detach(), sse.detach()
        A = [p_ for p_ in pol]
    sse_b = zero
    if which == 'u' and BXs.shape[0] > 0:
        sb = funcs['h'](BXs[:, 0, :]) if bgrid_first == T0 else funcs['g'](BXs[:, 0, :].unsqueeze(1)).reshape(-1)
        ub = R.u_net(th, config, BXs, sb)
        sse_b = ((ub - (gbar if pair_b else funcs['g'](BXs))) ** 2).sum()
        scal[3] = sse_b.detach()
    flat = lambda gs: torch.cat([(g_ if g_ is not None else torch.zeros_like(par[k])).reshape(-1) for k, g_ in zip(keys, gs)])  # noqa: E731
    P = sum(v_.numel() for v_ in par.values())
    if which == 'u':
        # ---- generator: ONE exchange of [J^T ubarA | J^T ubarB | partial sums]; an empty share contributes zeros
        pen = alpha * (sse / n + sse_b / (nb * Lb))
        gA = torch.autograd.grad(pen, [par[k] for k in keys], retain_graph=True, allow_unused=True) if pen.requires_grad else [None] * len(keys)
        gB = torch.autograd.grad(part_I, [par[k] for k in keys], allow_unused=True) if part_I.requires_grad else [None] * len(keys)
        packA = flat(gA) + (flat(A) if A is not None else torch.zeros(P, dtype=F64))
        pack = torch.cat([packA, flat(gB), scal]).detach().clone()
        world.all_reduce(pack)
        sc = pack[2 * P:]
        I = sc[0] - (V / n) * sc[7] * sc[8] if pair_i else sc[0]
        g = pack[:P] + (2.0 / I) * pack[P:2 * P]
        loss = torch.log(I ** 2) - torch.log(V * sc[1] / (n * L)) + alpha * (sc[2] / n + var_h + sc[3] / (nb * Lb) + var_g)
    else:
        # ---- discriminator: the partial sums first (the cotangent needs the global I and S), then the packed gradient
        world.all_reduce(scal)
        I = scal[0] - (V / n) * scal[7] * scal[8] if pair_i else scal[0]
        S = scal[1]
        sur = -(2.0 / I) * part_I + part_S / S
        gv = torch.autograd.grad(sur, [par[k] for k in keys], allow_unused=True) if sur.requires_grad else [None] * len(keys)
        g = (flat(gv) + (flat(A) if A is not None else torch.zeros(P, dtype=F64))).detach().clone()
        world.all_reduce(g)
        loss = -(torch.log(I ** 2) - torch.log(V * S / (n * L)))
    if carried is not None:
        g = g + carried
    # which parameters Adam skips: the field's, until a group of this sub-iteration has integrated the ODE -- from the GROUP's
    # shapes, not from what this rank happens to hold
    touched = touched or L > 1 or (nb > 0 and Lb > 1)
    grads, o = {}, 0
    for k in keys:
        m_ = par[k].numel()
        grads[k] = None if (which == 'u' and k in R.FIELD_KEYS and not touched) else g[o:o + m_].view_as(par[k]).clone()
        o += m_
    return float(loss), g, grads, touched, (Xs.shape[0], BXs.shape[0])


def _list_rank_worker(rank, size, port, out_dir, golden_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import configs.Ex4_1_funcs as F1
    from oracle import refspec as R
    from xnode_wan_pde_solver_amd import dist as xdist
    world, _ = xdist.init_from_env('gloo')
    config, setup, theta, phi, domain, triples = _tiny_groups(golden_dir)
    funcs = dict(h=F1.func_h, f=F1.func_f, g=F1.func_g, a=F1.func_a, b=F1.func_b, c=F1.func_c)
    ref = R.GroupLoop({k: v.clone() for k, v in theta.items()}, {k: v.clone() for k, v in phi.items()}, config, setup, domain, funcs)
    adam_u, adam_v, res = {}, {}, []
    for which in ('u', 'u', 'v'):
        outs = ref.sub_iteration(which, triples)                       # the unsharded loop of the reference (oracle restatement)
        carried, touched = None, False
        for k, triple in enumerate(triples):
            loss, carried, grads, touched, held = _rank_substep(which, theta, phi, config, setup, domain, funcs, triple, world, carried, touched)
            if which == 'u':
                theta = R.adam_update_sparse(theta, grads, adam_u, config['u_rate'])
            else:
                phi = R.adam_update_sparse(phi, grads, adam_v, config['v_rate'])
            want = outs[k]['grad']
            skipped = [kk for kk in grads if grads[kk] is None]
            assert skipped == [kk for kk in want if want[kk] is None], (which, k, skipped)
            gw = torch.cat([(want[kk] if want[kk] is not None else torch.zeros_like(grads[kk] if grads[kk] is not None else theta[kk])).reshape(-1)
                            for kk in want])
            res.append(dict(which=which, group=k, held=held, loss=(loss, float(outs[k]['loss'])),
                            gerr=float((carried - gw).abs().max()), gscale=float(gw.abs().max())))
        now, want_p = (theta, ref.theta) if which == 'u' else (phi, ref.phi)
        res.append(dict(which=which, perr=max(float((now[kk] - want_p[kk]).abs().max()) for kk in now)))
    torch.save(res, os.path.join(out_dir, 'lg%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('size', [4, 8])
def test_list_groups_smaller_than_the_world_keep_every_rank_in_step(tmp_path, golden_dir, size):
    """Groups of 5, 3, 3, 4, 1, 2 interior and 4, 3, 2, 1, 4, 3 boundary paths on 4 and 8 ranks: dist.World.shard_group leaves
    ranks with empty shares of the interior sample, of the boundary sample or of both; they contribute zeros to every exchange
    of the group's sub-step and apply the same update.  Two generator sub-iterations and one discriminator sub-iteration over
    the six groups (carried gradients, the pairwise single-slice group, Adam skipping the field until a group has integrated
    the ODE) must reproduce the unsharded GroupLoop -- the reference's own loop, pinned in test_oracle_golden -- on every rank."""
    mp.spawn(_list_rank_worker, args=(size, _free_port(), str(tmp_path), golden_dir), nprocs=size, join=True)
    ranks = [torch.load(tmp_path / ('lg%d.pt' % r)) for r in range(size)]
    held = [[e['held'] for e in r if 'held' in e][:6] for r in ranks]
    for k, (n, nb) in enumerate(zip(GROUP_SIZES, BDRY_SIZES)):
        assert sum(h[k][0] for h in held) == n and sum(h[k][1] for h in held) == nb      # every path on exactly one rank
    assert any(h[k] == (0, 0) for h in held for k in range(6))                           # a rank with NOTHING of a group
    assert any(h[k][0] == 0 and h[k][1] > 0 for h in held for k in range(6)) or size == 8
    for r in ranks:
        for e in r:
            if 'perr' in e:
                assert e['perr'] < 1e-9, e
            else:
                np.testing.assert_allclose(e['loss'][0], e['loss'][1], rtol=1e-8 if e['which'] == 'u' else 1e-6, err_msg=str(e))
                assert e['gerr'] <= 1e-6 * e['gscale'], e
        assert [e['loss'][0] for e in r if 'loss' in e] == [e['loss'][0] for e in ranks[0] if 'loss' in e]   # the same numbers on every rank


def test_schedule_selection_on_a_node_of_eight():
    """Which schedule a rank's shard takes is decided from its size alone (engine.Engine._compact / _narrow_ok: policy, no
    kernels): the headline batch over 8 ranks is 512 interior + 512 boundary paths per rank = 64 tiles -- the compact generator
    schedule with narrow-tile forward and x-only launches --, the unsharded batch keeps the wide schedule with 16-path waves;
    configs[2] over 8 ranks (2048 paths) is compact with 16-path waves."""
    from xnode_wan_pde_solver_amd.engine import Engine
    from xnode_wan_pde_solver_amd.dist import World

    class W(World):
        def __init__(self, rank, size):
            self.rank, self.size, self.group, self.comm = rank, size, None, None
    eng = Engine.__new__(Engine)
    eng.compact_tiles, eng.use_streams, eng.narrow, eng.narrow_set = 320, True, '1', 'fx'
    eng.narrow_tiles, eng.adjoint, eng.method = {'f': 192, 'x': 128, 'p': 64}, False, 1

    class G:
        pass

    def shard(n_glob, rank=0, size=8, d=20):
        lo, hi = W(rank, size).bounds(n_glob)
        g = G()
        g.N = g.Nb = hi - lo
        job = dict(xT=torch.empty(d, g.N), act=torch.empty(1))
        return g, job
    for rank in (0, 7):
        g, job = shard(4096, rank)
        assert g.N == 512 and eng._compact(g, True, True)
        assert eng._narrow_ok([job, job], alone=False, forward=True) and eng._narrow_ok([job], alone=False, params=False)
        assert not eng._narrow_ok([job, job, job], alone=False)              # sweeps with weight gradients stay on the duo waves
    g, job = shard(4096, 0, size=1)
    assert g.N == 4096 and not eng._compact(g, True, True) and not eng._narrow_ok([job, job], alone=False, forward=True)
    g, job = shard(16384, 3)
    assert g.N == 2048 and eng._compact(g, True, True) and not eng._narrow_ok([job, job], alone=False, forward=True)
    g, job = shard(4099, 0)
    assert g.N == 513 and eng._compact(g, True, True)


def _checksum_worker(rank, size, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from xnode_wan_pde_solver_amd import dist as xdist
    world, _ = xdist.init_from_env('gloo')
    g = torch.Generator().manual_seed(11)
    theta, phi = torch.randn(1983, generator=g, dtype=torch.float64), torch.randn(5851, generator=g, dtype=torch.float64)
    world.assert_in_step(theta, phi)                       # identical replicas: passes on every rank
    res = {'same': True}
    phi2 = phi.clone()
    if rank == size - 1:
        phi2[4097] = torch.nextafter(phi2[4097], torch.tensor(float('inf'), dtype=torch.float64))     # ONE ulp on ONE rank
    try:
        world.assert_in_step(theta, phi2)
        res['caught'] = False
    except RuntimeError as e:
        res['caught'] = 'drifted apart' in str(e)
    # swapped entries keep the plain sum: the position-weighted checksum still sees them
    theta2 = theta.clone()
    if rank == 0:
        theta2[[3, 7]] = theta2[[7, 3]]
    try:
        world.assert_in_step(theta2, phi)
        res['caught_swap'] = False
    except RuntimeError:
        res['caught_swap'] = True
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('size', [2, 4])
def test_replica_checksum_catches_a_one_ulp_drift_on_one_rank(tmp_path, size):
    """dist.World.assert_in_step (what train() ends with under a world): parameters are replicated and never broadcast, so a
    divergence can only be noticed by looking -- identical blobs pass, one ulp in one entry on one rank raises ON EVERY RANK
    (nobody is left waiting in the next collective), and so does a permutation that keeps the plain sum"""
    port = _free_port()
    mp.start_processes(_checksum_worker, args=(size, port, str(tmp_path)), nprocs=size, join=True, start_method='spawn')
    for r in range(size):
        res = torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r))
        assert res == {'same': True, 'caught': True, 'caught_swap': True}, (r, res)
