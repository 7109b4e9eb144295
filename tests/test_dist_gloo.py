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
def test_two_rank_gloo_exchange_reproduces_unsharded_step(tmp_path):
    size = 2
    mp.spawn(_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    res = [torch.load(tmp_path / ('rank%d.pt' % r)) for r in range(size)]
    assert res[0]['bounds'] == (0, 19) and res[1]['bounds'] == (19, 37)
    for r in res:
        for which in ('u', 'v'):
            loss, ref_loss, err, scale = r[which]
            np.testing.assert_allclose(loss, ref_loss, rtol=1e-6 if which == 'v' else 1e-9)
            assert err <= 2e-5 * scale, (which, err, scale)
    assert res[0]['u'][0] == res[1]['u'][0]               # every rank sees the same global loss


def test_bounds_cover_everything_once():
    from xnode_wan_pde_solver_amd.dist import World

    class W(World):
        def __init__(self, rank, size):
            self.rank, self.size, self.group = rank, size, None
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
def test_rank_local_sampling_draws_only_the_shard(tmp_path):
    """sampling.RankCubeLoader (solver.rank_local_sampling): every rank draws its share only; shares add up to the global
    counts, the time grids and the shared RNG stream stay common, the points are different draws on every rank, interior
    points are uniform in the cube and boundary points sit on faces in the global proportions"""
    size = 2
    mp.spawn(_sampling_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    r0, r1 = (torch.load(tmp_path / ('samp%d.pt' % r)) for r in range(size))
    assert torch.equal(r0[2]['after'], r1[2]['after'])
    for a, b in zip(r0[:2], r1[:2]):
        assert a['n'] + b['n'] == 1001 and a['nb'] + b['nb'] == 403 and abs(a['n'] - b['n']) <= 1
        assert torch.equal(a['t'], b['t'])
        assert a['xu'].shape == (a['n'], 5) and a['xb'].shape == (a['nb'], 5)
        assert not torch.equal(a['xu'][:100], b['xu'][:100]) and not torch.equal(a['xu'], a['xv'])
        for r in (a, b):
            assert float(r['xu'].abs().max()) <= 1.0 and abs(float(r['xu'].mean())) < 0.1
            on_face = (r['xb'].abs() == 1.0)
            assert bool(torch.all(on_face.sum(1) >= 1))
            per_axis = on_face.sum(0).double() / r['nb']
            assert float(per_axis.min()) > 0.08 and float(per_axis.max()) < 0.35      # ~ 1 / d each
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
