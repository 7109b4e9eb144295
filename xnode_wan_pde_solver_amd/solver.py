"""NODE_WAN_solver with the reference's constructor / attributes / train() behaviour (src/training.py:54-187), driving
the HIP engine instead of eager PyTorch + autograd.

What is kept (callers: main.py, example.ipynb, user `stop` hooks -- SURVEY.md section 8(b)):
  * NODE_WAN_solver(params, func_a, func_b, func_c, func_h, func_f, func_g, device, path, stop=None, func_u_sol=None, p=1)
  * attributes  u_net v_net config setup iterations domain n1 n2 optimizer_u optimizer_v best_l av_l params p func_*
  * train(report=False, report_it=10, show_plt=False): resample -> n1 generator sub-steps -> n2 discriminator sub-steps
    -> diagnostics, with the same side-effect files (losses_NODE_{d}.json, L2_NODE_{d}.json, Time_NODE_{d}.json,
    best_model_weights_NODE.pth) and the same `stop` hook protocol
  * state_dict keys of u_net / v_net (`module.` prefix, tied-layer aliases)
Differences, all deliberate:
  * `params` is read BY KEY (the reference slices the dict positionally, src/training.py:80-83, which breaks for the
    notebook's dict); `shape_param` defaults to [-1, 1] when absent; `domain` is looked up in a registry, not eval()'d
  * a CPU `device` is refused: there is no CPU path in this engine
  * `exit()` on the stop criterion can be turned into a normal return with `solver.exit_on_stop = False`
"""
import json
import os
import time
from itertools import product

import torch

from . import nets, sampling
from .sampling import HIP_HOST_LOCK
from ._lib import XnwanError
from .engine import Engine
from .options import EngineOptions
from .kernels import adam as _adam_kernel

CONFIG_KEYS = ['alpha', 'u_layers', 'u_hidden_dim', 'u_hidden_hidden_dim', 'v_layers', 'v_hidden_dim', 'n1', 'n2',
               'u_rate', 'v_rate', 'min_steps', 'adjoint', 'solver']
SETUP_KEYS = ['dim', 'N_t', 'N_r', 'N_b', 'T0', 'T', 'shape_param']


def split_params(params):
    missing = [k for k in CONFIG_KEYS + SETUP_KEYS[:-1] + ['iterations', 'domain'] if k not in params]
    if missing:
        raise KeyError('params is missing %s' % missing)
    config = {k: params[k] for k in CONFIG_KEYS}
    setup = {k: params[k] for k in SETUP_KEYS if k in params}
    setup.setdefault('shape_param', [-1, 1])     # the notebook's dict has none; Hypercube default (documented)
    return config, setup, int(params['iterations'])


def func_eval(X, BX, setup, y_output_u, func_a, func_b, func_c, func_h, func_f, func_g):
    """Tabulate the PDE data on a group (reference src/training.py:13-43): h[N], f[N,L], g[N_b,L], a[d,d,N,L],
    b[d,N,L], c[N,L,1] (attached to y_output_u).  Kept for user code; the engine itself never builds a[d,d,N,L]."""
    d = setup['dim']
    h, f, g = func_h(X[:, 0, :]), func_f(X), func_g(BX)
    c = func_c(X, y_output_u)
    a = torch.empty(d, d, X.shape[0], X.shape[1])
    for i, j in product(range(d), repeat=2):
        a[i, j] = func_a(X, i, j)
    b = torch.empty(d, X.shape[0], X.shape[1])
    for i in range(d):
        b[i] = func_b(X, i)
    dev = X.device
    return h.to(dev), f.to(dev), g.to(dev), a.to(dev), b.to(dev), c.to(dev)


def build_networks(config, setup, func_h, func_g, domain_cls):
    """Construct (u_net, v_net) on the host in the reference's order (src/training.py:88-100): a domain instance first
    (its time grid consumes N_t uniforms), then the XNODE, the test network, and the two Xavier passes."""
    s = setup
    domain = domain_cls(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
    xnode = nets.XNODE(config['u_hidden_dim'], 1, func_h, func_g, s, config['u_hidden_hidden_dim'], config['u_layers'],
                       domain, config['solver'], config['min_steps'], config['adjoint'])
    u_net = nets.PathParallel(xnode)
    v_net = nets.PathParallel(nets.TestNet(config, s))
    u_net.apply(nets.init_weights)
    v_net.apply(nets.init_weights)
    return u_net, v_net



class _JsonList(list):
    """A list whose JSON text is kept up to date as it grows.  The reference rewrites losses_NODE_{d}.json and
    Time_NODE_{d}.json with json.dump(whole list) after every sub-iteration (src/training.py:133-134,166-167): formatting
    k floats per write is milliseconds once k is in the thousands, more than the GPU work of the iteration it follows.
    The files written here are byte-identical to json.dump's; only the formatting is incremental."""

    def __init__(self, items=()):
        super().__init__(items)
        self._text = ', '.join(json.dumps(x) for x in self)

    def append(self, x):
        self._text += (', ' if len(self) else '') + json.dumps(x)
        super().append(x)

    def text(self):
        return '[' + self._text + ']'

    def write(self, path):
        _write_text(path, self.text())


def _run_all(jobs):
    for fn, args in jobs:
        fn(*args)


def _write_text(path, text):
    with open(path, 'w') as fh:
        fh.write(text)


class _Saver:
    """The side-effect files of train() (loss list, L2, times, best weights) written by ONE worker thread, in order -- a later
    write of a file overwrites an earlier one exactly as in a loop that writes them itself.  The main thread only hands over the
    finished text / host state dict: torch.save and the open() calls were a third of an outer iteration's host time, spent
    while the GPU had nothing queued.  A failed write (disk full, ...) surfaces at the next hand-over, not at the end."""

    def __init__(self, active=True):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=1) if active else None
        self.pending = []

    def submit(self, jobs):
        if self.pool is None or not jobs:
            return
        done = [f for f in self.pending if f.done()]
        self.pending = [f for f in self.pending if not f.done()]
        for f in done:
            f.result()                         # (raises here what the worker raised)
        self.pending.append(self.pool.submit(_run_all, jobs))

    def close(self, raising=True):
        """everything handed over is on disk when this returns; raising=False (another exception is on its way): best effort"""
        if self.pool is None:
            return
        self.pool.shutdown(wait=True)
        for f in self.pending:
            if raising:
                f.result()
        self.pending = []


class FusedAdam:
    """Stands where the reference has torch.optim.Adam (attributes optimizer_u / optimizer_v).  step() applies the
    fused HIP Adam kernel to the blob using the .grad of the parameters (for user code that went through autograd);
    the engine's own sub-steps call the same kernel with the slab gradients."""

    def __init__(self, blob, state, lr, on_step=None):
        self.blob, self.state, self.lr, self.on_step = blob, state, lr, on_step
        self.param_groups = [{'params': blob.params, 'lr': lr, 'betas': (0.9, 0.999), 'eps': 1e-8}]

    def zero_grad(self, set_to_none=True):
        for p in self.blob.params:
            p.grad = None

    def step(self):
        g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.blob.params])
        _adam_kernel(self.blob.data, None, self.state['m'], self.state['v'], self.state['step'],
                     self.param_groups[0]['lr'], gextraA=g.contiguous())
        if self.on_step is not None:
            self.on_step()

    def state_dict(self):
        sd = {'m': self.state['m'].clone(), 'v': self.state['v'].clone(), 'step': int(self.state['step'].item()), 'lr': self.lr}
        if 'lag' in self.state:            # (u_theta: how many updates skipped the field's parameters, engine.Engine.__init__)
            sd['lag'] = int(self.state['lag'].item())
        return sd

    def load_state_dict(self, sd):
        self.state['m'].copy_(sd['m'])
        self.state['v'].copy_(sd['v'])
        self.state['step'].fill_(sd['step'])
        if 'lag' in self.state:
            self.state['lag'].fill_(sd.get('lag', 0))


class NODE_WAN_solver:
    def __init__(self, params, func_a, func_b, func_c, func_h, func_f, func_g, device, path, stop=None,
                 func_u_sol=None, p=1, world=None, options=None):
        self.params = params
        # every switch of the engine and of train()'s loops (options.EngineOptions): filled ONCE from the XW_* environment here
        # unless the caller hands one in; passed down to the engine, printed by plan()
        opt = self.options = options if options is not None else EngineOptions.from_env()
        self.func_a, self.func_b, self.func_c = func_a, func_b, func_c
        self.func_h, self.func_f, self.func_g = func_h, func_f, func_g
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        if self.device.type != 'cuda':
            raise XnwanError("device %r: this engine only runs on an MI355X ('cuda' device of PyTorch-ROCm)" % (device,))
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.path, self.stop, self.func_u_sol, self.p = path, stop, func_u_sol, p
        self.world = world
        self.exit_on_stop = True
        self.tabulate_on_host = False     # False: h, f, g, w are tabulated on the GPU from the compact sample (fast; the
                                          # sample itself is still drawn with the host RNG, draw-for-draw like the reference);
                                          # True: tabulate on the host exactly like the reference's CPU path (bitwise h, f, g)
        self.device_sampling = False      # True: draw the cube samples with the device RNG (no seed parity, fastest)
        self.overlap_sampling = True      # draw the next host samples on a helper thread while the GPU works (same draws,
                                          # same order, nothing beyond the last iteration); off when a stop callback is set
        self.host_threads = 4             # intra-op CPU threads while train() runs (None: leave torch's setting alone).  The
                                          # host side of an outer iteration is a few small tensor ops (sampling, JSON,
                                          # torch.save); fanned out over every core of a big host they take 5x longer
                                          # and delay the kernel launches (measured: 67 -> 11 ms per outer iteration)
        self.rank_local_sampling = False  # several GPUs: True = every rank draws only its own share of the cube sample
                                          # (sampling.RankCubeLoader: no seed parity across rank counts); False = every
                                          # rank draws the global sample from the shared seed and keeps its slice
        self.pipeline = True              # single-group domains without a stop hook: nothing in an outer iteration waits for
                                          # the GPU -- losses, the diagnostic and the best weights of iteration k are read back,
                                          # written to the side-effect files and compared while iteration k + 1 runs (same
                                          # values, same files, one iteration later; everything is flushed before train()
                                          # returns).  False: every sub-iteration is synchronised like the reference's loop
        self.capture_refill = opt.capture_refill
                                          # the pipelined loop refills its group (path tensors, h, f, g, w, transposes: ~110 small
                                          # kernels) and evaluates the L^p diagnostic by replaying ONE captured graph each
                                          # (Engine.refill_compact, _l_norm_replayed); same arithmetic, same fallback rule as the
                                          # sub-step graphs.  False: eager launches
        self.defer_list_readback = opt.defer_list_readback
                                          # list domains (11-20 groups per sample) with the sampling thread: an outer iteration's
                                          # sub-steps are queued without a read-back and the NEXT sample's groups are loaded while
                                          # the GPU walks them; one read-back per outer iteration (_list_iteration_deferred).
                                          # False: every sub-iteration is synchronised like the reference's loop
        self.sampler_process = opt.sampler_process
                                          # the time-varying ball domains: the samples are drawn by a forked child process instead
                                          # of a helper thread (sampler_proc.py: the two sides of an outer iteration are ~2000 small
                                          # host operations each and two threads share the interpreter lock); same draws, the
                                          # generator states come back when train() ends
        self.reuse_test_net = True        # v, dv/dt, nabla_x v(t_0) are evaluated once per (phi, sample) and shared by the
                                          # sub-steps of an outer iteration -- bit-identical results (the reference
                                          # recomputes the same values); bench.py times the sub-steps WITHOUT it
        self.overlap_diagnostic = True    # pipelined loop: the diagnostic's graph beside the next sample's refill (two streams)
        self.check_replicas = True        # several GPUs: train() ends with a cross-rank checksum of theta and phi (dist.World.assert_in_step)
        self._group_cache = []
        self.config, self.setup, self.iterations = split_params(params)
        self.domain = sampling.resolve_domain(params['domain'])
        self.n1, self.n2 = self.config['n1'], self.config['n2']

        s = self.setup
        self.u_net, self.v_net = build_networks(self.config, s, func_h, func_g, self.domain)
        with torch.cuda.device(self.device):
            self.u_net.module.bind(self.device)
            self.v_net.module.bind(self.device)
            funcs = dict(a=func_a, b=func_b, c=func_c, h=func_h, f=func_f, g=func_g)
            self.engine = Engine(self.config, s, self.u_net.module, self.v_net.module, funcs, self.device, world=world, options=opt)
        self.optimizer_u = FusedAdam(self.u_net.module.blob, self.engine.adam_u, self.config['u_rate'])
        self.optimizer_v = FusedAdam(self.v_net.module.blob, self.engine.adam_v, self.config['v_rate'],
                                     on_step=self.engine.invalidate_test_net)
        self.best_l = float('inf')
        self.av_l = 0
        self.last_loss_u = self.last_loss_v = float('nan')

    def rebind(self):
        """re-alias the parameters to fresh blobs after the modules were moved or cast"""
        self.u_net.module.bind(self.device)
        self.v_net.module.bind(self.device)
        self.engine.theta, self.engine.phi = self.u_net.module.blob, self.v_net.module.blob
        self.optimizer_u.blob, self.optimizer_v.blob = self.engine.theta, self.engine.phi

    # --------------------------------------------------------------------------------------------------------------
    def _new_domain(self):
        s = self.setup
        return self.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])

    def _loader(self, domain, interior_only=False):
        # (rank-local sampling first: an interior_only request -- the diagnostic's sample -- must not turn a rank-local run
        #  into one where every rank draws and evaluates the GLOBAL sample; RankCubeLoader's share is what _l_norm_value's
        #  8-byte exchange combines)
        if self.world is not None and self.rank_local_sampling and hasattr(domain, 'interior_x'):
            return sampling.RankCubeLoader(self.setup['N_r'], self.setup['N_b'], domain, self.device, self.world.rank, self.world.size)
        if interior_only and not self.device_sampling and not self.tabulate_on_host:
            return sampling.Comb_loader(self.setup['N_r'], self.setup['N_b'], domain, self.device, interior_only=True)
        if self.device_sampling and hasattr(domain, 'device_sample'):
            return sampling.DeviceCubeLoader(self.setup['N_r'], self.setup['N_b'], domain, self.device)
        return sampling.Comb_loader(self.setup['N_r'], self.setup['N_b'], domain, self.device)

    def _groups(self, points):
        """(u, v, boundary) groups of a loader.  `tabulate_on_host=True` hands the engine the loader's HOST
        tensors, so that h, f, g, w are tabulated exactly like the reference's CPU path; False (default) builds the path tensors
        on the GPU from the compact sample and tabulates there."""
        self._grid_hint = None
        self._group_hints = None          # list domains: per group, what the engine would otherwise read back from the device
        self._rank_local = isinstance(points, sampling.RankCubeLoader)
        if not self.tabulate_on_host or self.device_sampling:
            packed = points.device_groups(self.device) if hasattr(points, 'device_groups') else None
            if packed is not None:        # a list sample: ONE upload, the groups are views of it
                groups, self._group_hints = packed
                return groups
            comp = points.compact() if hasattr(points, 'compact') else None
            if comp is not None:
                times, xu, xv, xb = comp
                if not times.is_cuda:
                    self._grid_hint = float(times[0])       # one shared grid: the engine need not read the tensors back
                td = self._up(times)
                mk = lambda x: sampling._paths(td, self._up(x))  # noqa: E731
                return [(mk(xu), mk(xv), mk(xb))]
            return list(points)
        if isinstance(points.interioru, list):
            n = min(len(points.interioru), len(points.boundary))      # same truncation as iterating the loader
            return [(points.interioru[i], points.interiorv[i], points.boundary[i]) for i in range(n)]
        return [(points.interioru, points.interiorv, points.boundary)]

    def _up(self, t):
        """host -> device; asynchronous when the source is page-locked (Comb_loader.pin)"""
        if t.is_cuda:
            return t
        if not t.is_pinned():
            return t.to(self.device)
        out = t.to(self.device, non_blocking=True)
        # (the staging slot is not overwritten before this copy has read it; the stream is named WITH its device index: without
        #  one torch asks whether CUDA is available on every call, 40 us)
        sampling._PIN_POOL.uploaded(t, torch.cuda.current_stream(self.device))
        return out

    def _shard(self, points):
        """(X, XV, BX, n_glob, nb_glob, grids) per group: this rank's contiguous share of every group (dist.py) with the path counts
        of the whole group and the time columns of its first interior / boundary path (Engine.load_group); identity -- the
        counts and grids None -- on one GPU and for the groups every rank computes in full (dist.World.replicated)"""
        if self.world is None:
            return [(du, dv, bd, None, None, None) for (du, dv, bd) in points]
        if self._rank_local:                       # the loader already drew this rank's share only
            return [(du, dv, bd, self.setup['N_r'], self.setup['N_b'], None) for (du, dv, bd) in points]
        out = []
        for du, dv, bd in points:
            # (replicas must compute bit-identical inputs: with tabulate_on_host the callables run on the concatenation of THIS
            #  rank's shares, where a replicated group sits at another offset on every rank -- a vectorised body / scalar tail
            #  may then differ in the last bit and nothing would ever resynchronise theta and phi.  Shard such runs instead.)
            if len(points) > 1 and not self.tabulate_on_host and self.world.replicated(du.shape[0], bd.shape[0]):
                out.append((du, dv, bd, None, None, None))
            else:
                out.append(self.world.shard_group(du, dv, bd) + ((du[0, :, 0], bd[0, :, 0]),))
        return out

    def _l_norm(self, points, volume, as_tensor=False):
        """the L^p diagnostic of src/training.py:167; as_tensor: a 0-dim float64 device tensor, nothing read back"""
        if as_tensor:
            val = self._l_norm_value(points, volume)
            return (val if torch.is_tensor(val) else torch.tensor(val)).to(device=self.device, dtype=torch.float64).reshape(())
        val = self._l_norm_value(points, volume)
        return val.item() if torch.is_tensor(val) else val

    def _l_norm_replayed(self, G, points, domain):
        """_l_norm(points, V, as_tensor=True) for a compact cube sample, as ONE graph replay (held with the group's sub-step
        graphs, same capture / fallback rule: Engine._run): the sample goes into two static buffers, the path tensor, the exact
        solution, u_theta (the stepper's forward kernel) and the mean are replayed.  Anything else takes the ordinary way."""
        comp = points.compact() if hasattr(points, 'compact') else None
        eng = self.engine
        if (comp is None or comp[0].is_cuda or self.func_u_sol is None or isinstance(points, sampling.RankCubeLoader)
                or not eng.use_graphs or self.tabulate_on_host):
            # (several GPUs with the shared seed: every rank evaluates the diagnostic on the global sample, like _l_norm_value)
            return self._l_norm(points, domain.V(), as_tensor=True)
        from utils.auxillary_funcs import L_norm
        times, xu = comp[0], comp[1]
        st = G.__dict__.get('_diag_in')
        if st is None or st[0].shape != times.shape or st[1].shape != xu.shape:
            st = G._diag_in = [torch.empty(tuple(c.shape), dtype=c.dtype, device=self.device) for c in (times, xu)] + [
                torch.zeros((), dtype=torch.float64, device=self.device)]
            G.graphs = {k: v for k, v in G.graphs.items() if not k.startswith('diag')}
        st[0].copy_(times, non_blocking=True)
        st[1].copy_(xu, non_blocking=True)
        sampling._PIN_POOL.uploaded_all((times, xu), torch.cuda.current_stream(self.device))
        at_T0 = float(times[0]) == self.setup['T0']
        volume, n_r, p = domain.V(), self.setup['N_r'], self.p

        def body(_G):
            X = sampling._paths(st[0], st[1])
            val = L_norm(X, lambda x: self.u_net(x, starts_at_T0=at_T0), p, self.func_u_sol, volume, n_r)
            st[2].copy_(val.to(torch.float64).reshape(()))
        key = 'diag_%r_%r_%r' % (at_T0, float(volume), p)
        if key not in G.graphs and sum(1 for k_ in G.graphs if k_.startswith('diag')) >= eng.refill_variants:
            G.graphs[key] = False        # (as Engine.refill_compact: a volume that changes every sample must not capture a graph per iteration)
        eng._run(G, key, body, scratch=2)         # (a scratch pool of its own: replayed beside the next sample's refill, pool 0)
        return st[2]

    def _l_norm_value(self, points, volume):
        from utils.auxillary_funcs import L_norm
        if self.func_u_sol is None:
            return float('nan')
        if isinstance(points, sampling.RankCubeLoader):
            # every rank holds its own paths: combine the p-th power sums with the global 1/N_r (one 8-byte exchange)
            comp = points.compact()
            X = sampling._paths(comp[0].to(self.device), comp[1].to(self.device))
            local = L_norm(X, self.u_net, self.p, self.func_u_sol, volume, points.n_local)
            part = (local.double() ** self.p / volume * (points.n_local / self.setup['N_r'])).reshape(1).to(self.device).contiguous()
            self.world.all_reduce(part)
            return (volume * part[0]) ** (1.0 / self.p)
        if not self.tabulate_on_host or self.device_sampling:
            comp = points.compact() if hasattr(points, 'compact') else None
            if comp is not None:          # diagnostic entirely on the device, nothing read back (the grid's first time is
                X = sampling._paths(self._up(comp[0]), self._up(comp[1]))   # known on the host: no sync in forward)
                at_T0 = (float(comp[0][0]) == self.setup['T0']) if not comp[0].is_cuda else None
                u_fn = lambda x: self.u_net(x, starts_at_T0=at_T0)   # noqa: E731
                return L_norm(X, u_fn, self.p, self.func_u_sol, volume, self.setup['N_r'])
            groups = points.interioru
            # (several ranks: every rank evaluates the diagnostic on the whole sample, as _l_norm_value does for the cube)
            lean = self._l_norm_groups(points, volume) if (isinstance(groups, list) and groups) else None
            if lean is not None:
                return lean
            if isinstance(groups, list) and groups and not groups[0].is_cuda:
                # list domain: the groups go to the device, func_u_sol is evaluated ONCE on all their points, u_theta group by
                # group with the start kind read from the host copy (no read-back between the launches)
                d1 = self.setup['dim'] + 1
                at0 = iter([float(g[0, 0, 0].detach()) == self.setup['T0'] for g in groups])
                Xs = [g.detach().to(self.device) for g in groups]
                sol = self.func_u_sol(torch.cat([x.reshape(-1, 1, d1) for x in Xs], 0)).reshape(-1)
                sols = iter([s_.view(x.shape[0], x.shape[1]) for s_, x in zip(sol.split([x.shape[0] * x.shape[1] for x in Xs]), Xs)])
                return L_norm(Xs, lambda x: self.u_net(x, starts_at_T0=next(at0)), self.p, lambda x: next(sols), volume, self.setup['N_r'])
        return L_norm(points.interioru, self.u_net, self.p, self.func_u_sol, volume, self.setup['N_r'])

    def _l_norm_groups(self, points, volume):
        """L_norm over the groups of a list-domain sample with the per-group overheads taken out -- the sample arrives in one
        upload (Comb_loader.device_groups), the exact solution and the start values h / g are evaluated ONCE on all groups'
        points (the callables are functions of the point: the engine checks that on its first sample, Engine._batch_tab), and
        u_theta is the stepper's forward kernel called directly per group (no module call, autograd Function or operator
        dispatch: 20 groups x 0.3 ms were 5 of an hourglass iteration's 29 ms).  utils.L_norm itself -- the reference's
        arithmetic, including its [N, N] table on a single-slice group -- is called as it is.  None: not applicable."""
        from utils.auxillary_funcs import L_norm
        from . import kernels as KN
        eng, net = self.engine, self.u_net.module
        packed = points.device_interior(self.device) if hasattr(points, 'device_interior') else None
        if packed is None or getattr(eng, '_batch_tab', True) is False or getattr(eng, '_batch_tab_checked', False) is False:
            return None
        Xs, hints = packed
        d1, T0 = self.setup['dim'] + 1, self.setup['T0']
        at0 = [h['t0'] == T0 for h in hints]
        with torch.no_grad():
            sol = self.func_u_sol(torch.cat([x.reshape(-1, 1, d1) for x in Xs], 0)).reshape(-1)
            sols = iter([s_.view(x.shape[0], x.shape[1]) for s_, x in zip(sol.split([x.shape[0] * x.shape[1] for x in Xs]), Xs)])
            # start values: h on the first points of the groups that start at T0, g on those that start on the moving boundary
            P0 = torch.cat([x[:, 0, :] for x in Xs], 0)
            n = [x.shape[0] for x in Xs]
            # (groups are contiguous row ranges of P0: slices, not an index tensor -- see Engine.tabulate_sample)
            rows = P0.split(n)
            rows_h, rows_g = [r for r, a in zip(rows, at0) if a], [r for r, a in zip(rows, at0) if not a]
            hval = self.func_h(torch.cat(rows_h, 0)).reshape(-1).double() if rows_h else None
            gval = self.func_g(torch.cat(rows_g, 0).unsqueeze(1)).reshape(-1).double() if rows_g else None
            it_h = iter(hval.split([r.shape[0] for r in rows_h])) if hval is not None else None
            it_g = iter(gval.split([r.shape[0] for r in rows_g])) if gval is not None else None
            start = torch.cat([next(it_h) if a else next(it_g) for a in at0], 0)
            starts = iter(zip(start.split(n), at0))
            net.blob.check_alias()

            def u_fn(x):
                s_k, a_k = next(starts)
                u, _ = KN.ode_fwd(x[:, 0, 1:].t().contiguous(), x[0, :, 0].contiguous(), s_k.contiguous(), net.blob.data, net.method,
                                  net.kdims[0], net.kdims[1], net.num_layers, want_Y=False)
                out = u.t().unsqueeze(2).contiguous()               # [N, L, 1] like xnwan::xnode_forward
                return out[:, 0, :] if (x.shape[1] == 1 and a_k) else out   # (src/model.py:89-91: [N, 1] on a single slice at T0)
            return L_norm(Xs, u_fn, self.p, lambda x: next(sols), volume, self.setup['N_r'])

    def plan(self, report=False):
        """Which of train()'s loops and helpers THIS solver would take, as a dict -- the choice depends on the domain, a `stop`
        hook, `report`, several ranks and a handful of attributes / XW_* switches, and the loops differ by a factor of three
        in host time (same results: tests/test_gpu_engine.py compares them bit for bit).  XW_SHOW_PLAN=1 prints it once per
        train() call."""
        eng = self.engine
        cube = hasattr(self._new_domain_probe(), 'interior_x')
        listy = isinstance(self.domain, type) and issubclass(self.domain, sampling._NSphereBase)
        overlap = self.overlap_sampling and self.stop is None and not self.device_sampling
        proc = bool(self.sampler_process and overlap and self.defer_list_readback and not self.tabulate_on_host and listy and hasattr(os, 'fork'))
        if self.pipeline and self.stop is None and not report and cube:
            loop = 'pipelined (nothing in an outer iteration waits for the GPU; files one iteration behind, flushed at the end)'
        elif not cube and overlap and self.defer_list_readback:
            loop = 'list domain, one read-back per outer iteration (next sample loaded behind the queued sub-steps)'
        else:
            loop = 'synchronous (a read-back after every sub-iteration, like the reference)'
        return {
            'loop': loop,
            'sampling': 'device RNG' if self.device_sampling else ('forked sampling process' if proc else 'helper thread' if overlap else
                                                                  'in the loop (a stop hook may draw random numbers itself)'),
            'tabulation': 'host (bitwise the reference\'s CPU values)' if self.tabulate_on_host else 'device',
            'refill': 'one graph replay per sample' if (cube and self.capture_refill and eng.use_graphs and not self.tabulate_on_host) else
                      ('one packed upload + one gather launch per sample' if (not cube and eng.packed_load and not self.tabulate_on_host) else 'group by group'),
            'sub_steps': 'captured HIP graphs' if (cube and eng.use_graphs) else ('one C call per group sub-step (xw_substep_*)' if eng.use_runner and
                                                                                 eng.structure.c_kappa is not None else 'launch by launch'),
            'ranks': 1 if self.world is None else self.world.size,
            'exchange': None if self.world is None else ('xw_allreduce (RCCL) on the stream: inside the sub-step graphs / the group runner'
                                                         if self.world.capturable else 'torch.distributed, staged through the host'),
            'small_groups': None if (self.world is None or cube) else ('replicated below %d paths per rank' % self.world.replicate_below
                                                                       if self.world.replicate_below > 0 else 'sharded (empty shares)'),
            'test_net_reuse_inside_an_outer_iteration': bool(self.reuse_test_net),
            'coefficients': eng.structure.describe(),
            'options_not_at_their_defaults': self.options.non_default(),
        }

    def train(self, report=False, report_it=10, show_plt=False):
        if self.options.show_plan and self._is_main():
            print('train() plan: ' + json.dumps(self.plan(report)))
        threads = torch.get_num_threads()
        if self.host_threads:
            torch.set_num_threads(min(threads, int(self.host_threads)))
        self.engine.reuse_test_net = bool(self.reuse_test_net) or self.engine.reuse_test_net
        try:
            out = self._train(report, report_it, show_plt)
            if self.world is not None and self.check_replicas:
                # parameters are replicated, never broadcast: look once per call that the ranks still hold the same bits
                self.world.assert_in_step(self.engine.theta.data, self.engine.phi.data)
            return out
        finally:
            torch.set_num_threads(threads)
            try:                                   # (the diagnostics' per-sample cache of func_u_sol: nothing outlives a run)
                from utils.auxillary_funcs import clear_exact_cache
                clear_exact_cache()
            except ImportError:
                pass

    def _train(self, report, report_it, show_plt):
        past_losses = _JsonList()
        times = _JsonList([time.time()])
        # The host draws (CPU generator, for the reference's seeds) cost about as much as the GPU work of an iteration.
        # They depend on nothing the steps compute, so one helper thread draws the post-step diagnostic sample of this
        # iteration and the domain + sample of the next one WHILE the main thread waits for the GPU -- in the reference's
        # order, and nothing beyond the last iteration, so the generator ends where the reference's does.  A user stop
        # callback may draw random numbers itself between those calls: then everything stays in line.
        pool = self._sampling_process()
        if pool is not None:
            try:
                pool.begin()
            except Exception as e:       # (held by another solver that is training, or the child is gone: the helper thread draws the same numbers)
                import warnings
                warnings.warn('the sampling process is not available (%s): drawing on a helper thread' % e, RuntimeWarning)
                pool = None
        if pool is not None:
            pass
        elif self.overlap_sampling and self.stop is None and not self.device_sampling:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1, initializer=torch.set_num_threads, initargs=(torch.get_num_threads(),))

        self._sampler_seconds = 0.0       # time the helper thread spent drawing (tools/train_cfg5.py: what bounds the ball domains)

        def draw_ahead(domain, last):
            t_ = time.perf_counter()
            pin = lambda ld: ld.pin() if hasattr(ld, 'pin') else ld     # noqa: E731  (page-locked: asynchronous uploads)
            after = pin(self._loader(domain, interior_only=True))     # (the diagnostic's sample: only its interior is read)
            if last:
                self._sampler_seconds += time.perf_counter() - t_
                return after, None, None
            nxt = self._new_domain()
            out = after, nxt, pin(self._loader(nxt))
            self._sampler_seconds += time.perf_counter() - t_
            return out

        try:
            if self.pipeline and self.stop is None and not report and hasattr(self._new_domain_probe(), 'interior_x'):
                return self._iterate_pipelined(past_losses, times, pool, draw_ahead)
            return self._iterate(report, report_it, show_plt, past_losses, times, pool, draw_ahead)
        finally:
            if pool is not None:
                pool.shutdown(wait=True)

    def _sampling_process(self):
        """the forked sampler of sampler_proc.py when this run can use it, else None (the helper thread, or no overlap at all)"""
        if not (self.sampler_process and self.overlap_sampling and self.defer_list_readback and self.stop is None
                and not self.device_sampling and not self.tabulate_on_host and isinstance(self.domain, type)
                and issubclass(self.domain, sampling._NSphereBase) and hasattr(os, 'fork')):
            return None
        s = self.setup
        key = (self.domain, s['N_r'], s['N_b'], s['dim'], s['N_t'], s['T0'], s['T'], repr(s['shape_param']))
        held = getattr(self, '_sampler_proc', None)
        if held is not None and held[0] == key and held[1].proc.is_alive():
            return held[1]
        if held is not None:
            held[1].close()
        try:
            from .sampler_proc import SamplerProcess
            sp = SamplerProcess(self.domain, dict(s), s['N_r'], s['N_b'])
        except Exception as e:          # (no shared memory, no fork: the helper thread draws the same numbers)
            import warnings
            warnings.warn('the sampling process could not be started (%s): drawing on a helper thread' % e, RuntimeWarning)
            self.sampler_process = False
            return None
        self._sampler_proc = (key, sp)
        return sp

    def _new_domain_probe(self):
        """the domain CLASS's compact-draw capability, without constructing an instance (construction draws random numbers)"""
        return self.domain

    # --------------------------------------------------------------------------------------------------------------
    # the pipelined loop: same work, same values, same files as _iterate below -- but the host never waits for the GPU
    # inside an outer iteration.  What the reference reads back after every sub-iteration (loss_u.item() for the loss
    # list and the best-weights rule, L_norm(...).item()) is copied into a small device ring together with a snapshot of
    # theta per generator sub-iteration; a side stream carries ring row k to pinned host memory behind an event, and the
    # host processes row k - 1 (JSON files, best_l, torch.save of the snapshot) while the GPU works on iteration k.
    # --------------------------------------------------------------------------------------------------------------
    def _iterate_pipelined(self, past_losses, times, pool, draw_ahead):
        d, dev, eng = self.setup['dim'], self.device, self.engine
        R, n1, n2 = 4, self.n1, self.n2
        # (ring, page-locked mirrors, events and the read-back stream live as long as the solver: allocating them -- two
        #  hipHostMalloc among them -- was 11 ms of every train() call, 0.45 ms per outer iteration of bench.py's 25-iteration calls)
        keep = getattr(self, '_pipe_buffers', None)
        if keep is None or keep[0] != (R, n1, eng.Pu, dev):
            ring = torch.zeros(R, n1 + 2, dtype=torch.float64, device=dev)          # loss_u x n1, loss_v, L2
            snaps = torch.zeros(R, n1, eng.Pu, dtype=torch.float64, device=dev)     # theta after every generator sub-iteration
            with HIP_HOST_LOCK:
                host = torch.zeros(R, n1 + 2, dtype=torch.float64).pin_memory()
                snaps_host = torch.zeros(R, n1, eng.Pu, dtype=torch.float64).pin_memory()   # (rides along with the ring row: 26 KB)
            keep = self._pipe_buffers = ((R, n1, eng.Pu, dev), ring, snaps, host, snaps_host,
                                         [torch.cuda.Event() for _ in range(R)], [torch.cuda.Event() for _ in range(R)],
                                         torch.cuda.Stream(device=dev))
        _, ring, snaps, host, snaps_host, done, filled, rb = keep
        # The files of an iteration (loss list, L2, times, best weights) are written by ONE worker thread, in order (a later write
        # of a file overwrites an earlier one exactly as in the synchronous loop); the main thread only hands it the finished text /
        # state dict -- torch.save and four open() calls were 1 ms of its 3 ms per iteration
        saver = _Saver(self._is_main())
        keys = self._state_dict_layout()

        def process(k):
            """host side of iteration k (its GPU work has finished or is finishing): files, best weights.  Runs while the GPU
            works on iteration k + 1, so Time_NODE_*.json holds the times the host WROTE an iteration's files (one iteration
            behind the GPU, same spacing), and the loss / L2 / best-weights files trail the GPU by one iteration; all of them
            are flushed before train() returns or raises (the `finally` below)."""
            r = k % R
            done[r].synchronize()
            row = host[r].tolist()
            jobs = []                      # this iteration's files, in the order the synchronous loop writes them: ONE hand-over
            for i in range(n1):
                self.av_l = row[i]
                self.last_loss_u = row[i]
                past_losses.append(self.av_l)
                jobs.append((_write_text, ('losses_NODE_' + str(d) + '.json', past_losses.text())))
                if self.av_l < self.best_l:
                    if self._is_main():
                        sd = self._state_dict_from(snaps_host[r, i].clone(), keys)   # (a copy of its own: Blob.split views the STORAGE from offset 0, and row r is reused R iterations later)
                        jobs.append((torch.save, (sd, 'best_model_weights_NODE.pth')))
                    self.best_l = self.av_l
            self.last_loss_v = row[n1]
            times.append(time.time())
            jobs.append((_write_text, ('L2_NODE_' + str(d) + '.json', json.dumps([row[n1 + 1]]))))
            jobs.append((_write_text, ('Time_NODE_' + str(d) + '.json', times.text())))
            saver.submit(jobs)

        # (Refilling a SECOND group for iteration k + 1 on a side stream beside the sub-steps of iteration k, and the diagnostic
        #  beside the discriminator sub-step, was built and measured: with torch's side stream on a hardware queue of its own
        #  (GPU_MAX_HW_QUEUES=8) an outer iteration takes 4.1-6.7 ms instead of 2.0-2.3 -- every one of the ~110 small dependent
        #  kernels of a refill waits for a CU with free registers and LDS next to resident test-network and stepper blocks that
        #  hold them for 0.2-0.3 ms; on the default four queues the side stream shares the main stream's queue and nothing
        #  overlaps.  One group, one stream.)
        main = torch.cuda.current_stream(dev)
        group = self._group_cache[0] if len(self._group_cache) == 1 else None

        def compact_of(pts):
            """(compact sample of THIS rank, global path counts) when the group can be refilled by graph replay, else None"""
            comp = pts.compact() if hasattr(pts, 'compact') else None
            if comp is None or not self.capture_refill or not eng.use_graphs or comp[0].is_cuda or self.tabulate_on_host:
                return None
            if self.world is None:
                return comp, None, None
            if isinstance(pts, sampling.RankCubeLoader):            # the loader already drew this rank's share only
                return comp, self.setup['N_r'], self.setup['N_b']
            times, xu, xv, xb = comp                                  # every rank drew the global sample: its contiguous slice
            (lo, hi), (blo, bhi) = self.world.bounds(xu.shape[0]), self.world.bounds(xb.shape[0])
            if hi - lo == 0 or bhi - blo == 0:
                return None
            return (times, xu[lo:hi], xv[lo:hi], xb[blo:bhi]), xu.shape[0], xb.shape[0]

        # (host seconds per phase of the loop body, summed over the run: tools/train_phases.py prints them)
        phase = self._phase_seconds = dict.fromkeys(('fill', 'substeps', 'sampler_wait', 'diagnostic', 'ring', 'process'), 0.0)
        clock = time.perf_counter

        def lap(name, t):
            now = clock()
            phase[name] += now - t
            return now

        nxt_domain = nxt_points = ahead = None
        # The diagnostic of iteration k (a chain of ~45 small launches around one forward pass of the stepper, on a FRESH sample)
        # and the refill of the group for iteration k + 1 (~85 small launches) do not depend on each other and each leaves the
        # chip nearly empty: the diagnostic's graph is replayed on a stream of its own while the main stream replays the refill,
        # and the main stream waits for it before the first sub-step (beside the persistent blocks of the test network its
        # small launches would starve).  Same values, same files; train() 1.8x -> 1.6x ms per outer iteration at the headline size.
        side = self.__dict__.get('_diag_stream')
        if side is None:
            side = self._diag_stream = torch.cuda.Stream(device=dev)
        prefilled = False
        failing = False
        issued = processed = 0          # iterations whose ring row is on its way to the host / whose files have been written
        last = self.iterations - 1
        with torch.cuda.device(dev):
          try:
            for k in range(self.iterations):
                domain = nxt_domain if nxt_domain is not None else self._new_domain()
                points = nxt_points if nxt_points is not None else self._loader(domain)
                nxt_domain = nxt_points = None
                if ahead is None and pool is not None:     # (later ones are submitted the moment the previous result is taken)
                    ahead = pool.submit(draw_ahead, domain, k == last)
                tick = clock()
                old = group
                comp = compact_of(points)
                if prefilled:
                    G, prefilled = old, False             # (refilled at the end of the previous iteration, beside its diagnostic)
                elif old is not None and comp is not None:
                    # every iteration after the first: the sample into static buffers, ONE graph replay fills the group
                    G = eng.refill_compact(old, comp[0], domain, comp[1], comp[2])
                else:
                    (du, dv, bd, ng, nbg, grids), = self._shard(self._groups(points))
                    G = eng.load_group(du, dv, bd, domain, ng, nbg, into=old, shared_grid_t0=self._grid_hint, grids=grids)
                group = G
                self._group_cache = [G]
                G.persistent = True
                r = k % R
                if k >= R:
                    done[r].synchronize()                 # (row r was processed R - 1 iterations ago; its copy is long done)
                tick = lap('fill', tick)
                for i in range(n1):
                    eng.begin_substep('u', False)
                    eng.generator_step(G)
                    ring[r, i].copy_(eng.loss_u())
                    snaps[r, i].copy_(eng.theta.data)
                for _ in range(n2):
                    eng.begin_substep('v', False)
                    eng.discriminator_step(G)
                ring[r, n1].copy_(eng.loss_v())
                tick = lap('substeps', tick)
                if ahead is not None:
                    points_after, nxt_domain, nxt_points = ahead.result()
                    ahead = pool.submit(draw_ahead, nxt_domain, k + 1 == last) if k < last else None
                else:
                    points_after = self._loader(domain)
                tick = lap('sampler_wait', tick)
                comp_n = compact_of(nxt_points) if (nxt_points is not None and self.overlap_diagnostic and self.capture_refill
                                                    and self.world is None and eng.use_streams) else None
                if comp_n is not None:
                    e_sub = main.record_event()
                    torch.cuda.set_stream(side)
                    try:
                        side.wait_event(e_sub)
                        diag = self._l_norm_replayed(G, points_after, domain)
                        ring[r, n1 + 1].copy_(diag)
                        e_diag = side.record_event()
                    finally:
                        torch.cuda.set_stream(main)
                    tick = lap('diagnostic', tick)
                    # ... and beside it the NEXT sample into the group (its sub-steps are all queued in front of this replay)
                    group = eng.refill_compact(G, comp_n[0], nxt_domain, comp_n[1], comp_n[2])
                    prefilled = True
                    main.wait_event(e_diag)
                    tick = lap('fill', tick)
                else:
                    diag = (self._l_norm_replayed(G, points_after, domain) if self.capture_refill
                            else self._l_norm(points_after, domain.V(), as_tensor=True))
                    tick = lap('diagnostic', tick)
                    ring[r, n1 + 1].copy_(diag)
                filled[r].record(main)
                torch.cuda.set_stream(rb)
                try:
                    rb.wait_event(filled[r])
                    host[r].copy_(ring[r], non_blocking=True)
                    snaps_host[r].copy_(snaps[r], non_blocking=True)
                    done[r].record(rb)
                finally:
                    torch.cuda.set_stream(main)
                issued = k + 1
                tick = lap('ring', tick)
                if k > 0:
                    processed = k                         # (before the call: see the flush below)
                    process(k - 1)
                lap('process', tick)
          except BaseException:
            failing = True
            raise
          finally:
            # the host side runs one iteration behind the GPU: whatever has been computed is written out before train() returns
            # OR raises (an exception / KeyboardInterrupt inside the loop must not lose the last iteration's losses and best weights)
            try:
                while processed < issued:
                    processed += 1                        # (advanced first: an iteration whose processing raised is not processed twice)
                    process(processed - 1)
            finally:
                saver.close(raising=not failing)          # every best-weights file is on disk before train() returns or raises
        return past_losses

    def _state_dict_layout(self):
        """[(state_dict key, index of its parameter in the blob's parameter list)]: tied layers appear under several keys"""
        blob = self.u_net.module.blob
        where = {p.data_ptr(): i for i, p in enumerate(blob.params)}
        return [(k_, where[v.data_ptr()]) for k_, v in self.u_net.state_dict().items()]

    def _state_dict_from(self, flat_host, keys):
        """u_net.state_dict() as it was when `flat_host` (a host copy of the parameter blob) was taken"""
        from collections import OrderedDict
        parts = self.u_net.module.blob.split(flat_host)
        sd = OrderedDict()
        for k_, i in keys:
            sd[k_] = parts[i].clone()
        return sd

    def _prepare_groups(self, points, domain, pairs_last=False):
        """the groups of a sample, loaded into the engine (the cached Group objects are refilled)"""
        eng = self.engine
        comp = points.compact() if hasattr(points, 'compact') else None
        if (comp is not None and len(self._group_cache) == 1 and self._group_cache[0] is not None and self.capture_refill
                and self.world is None and eng.use_graphs and not comp[0].is_cuda and not self.tabulate_on_host):
            # the cube after its first sample: static inputs + one graph replay (Engine.refill_compact)
            G = eng.refill_compact(self._group_cache[0], comp, domain)
            G._refill_points = points            # (whose sample the group's static input buffers hold: _stop_agreed)
            return [G]
        shards = self._shard(self._groups(points))
        if len(self._group_cache) != len(shards):     # (the number of groups varies from sample to sample: keep the ones that stay)
            self._group_cache = self._group_cache[:len(shards)] + [None] * (len(shards) - len(self._group_cache))
        # list domains: the callables are evaluated once for all groups of the sample; the structure guard runs on the
        # largest group of the sample (all groups are slices of the same draw)
        # (the loader's facts about a group -- first times, shared time column, same grid -- hold for every rank's share of it)
        hints = self._group_hints if self._group_hints is not None else [None] * len(shards)
        # (several ranks: a sample is tabulated on this rank's shares; the start kinds come from the whole groups' first paths)
        grids = [sh[5] for sh in shards]
        tabs = eng.tabulate_sample([sh[:3] for sh in shards], domain, hints=hints, grids=grids) if len(shards) > 1 else [None]
        big = max(range(len(shards)), key=lambda i: shards[i][0].shape[0] * shards[i][0].shape[1])
        order = list(range(len(shards)))
        if pairs_last:
            # (a single-slice group at T0 reads five sums back, Engine.load_group: behind queued sub-steps that read-back waits for
            #  all of them, so those groups are loaded after everything that does not wait)
            T0 = self.setup['T0']
            waits = lambda i: (shards[i][0].shape[1] == 1 or shards[i][2].shape[1] == 1) and hints[i] is not None and (  # noqa: E731
                hints[i]['t0'] == T0 or hints[i]['tb0'] == T0)
            order.sort(key=lambda i: bool(waits(i)) or hints[i] is None)
        if len(shards) > 1 and eng.packed_load and tabs[0] is not None:
            packed = eng.load_groups_packed(shards, hints, domain, self._group_cache, big)      # (one gather launch for all groups)
            if packed is not None:
                return packed
        groups = [None] * len(shards)
        for i in order:
            du, dv, bd, ng, nbg, grids = shards[i]
            groups[i] = eng.load_group(du, dv, bd, domain, ng, nbg, into=self._group_cache[i], shared_grid_t0=self._grid_hint, tab=tabs[i],
                                       verify=(i == big), hints=hints[i], grids=grids)
        return groups

    def _list_iteration_deferred(self, groups, domain, ahead, pool, draw_ahead, k, last, past_losses, times):
        """One outer iteration on a list domain (11-20 groups per sample) whose host work runs BESIDE its GPU work: the sub-steps
        of all groups are queued without a read-back (per-group losses and theta after every generator sub-iteration stay on the
        device), then -- while the GPU walks that chain of ~60 dependent sub-steps -- the host takes the sampling thread's
        result, queues the diagnostic and loads the NEXT sample's groups (callables tabulated, Group objects refilled: same
        stream, so behind the sub-steps that still read the old contents), and only then reads everything back at once.  The
        synchronous order (_iterate) has the GPU idle during the 10 ms of loading and the host idle during the 13 ms of sub-steps.
        Same values, same files, same order of writes; needs the sampling thread or process (no stop hook).  Several ranks walk
        it in lockstep: every exchange of a sub-step is a call on the stream (RCCL) or staged through the host (gloo rehearsal)."""
        eng, d, n1, n2 = self.engine, self.setup['dim'], self.n1, self.n2
        # (host seconds per phase, summed over the run: tools/train_cfg5.py prints them)
        phase = self.__dict__.setdefault('_list_phase_seconds', dict.fromkeys(('substeps', 'sampler_wait', 'diagnostic', 'load_next', 'read_back', 'files'), 0.0))
        clock = time.perf_counter

        def lap(name, t):
            now = clock()
            phase[name] += now - t
            return now
        tick = clock()
        lu, snaps = [], []
        for _ in range(n1):
            eng.begin_substep('u', True)
            for G in groups:
                eng.generator_step(G)
                lu.append(eng.loss_u().clone())
            snaps.append(eng.theta.data.clone())
        for _ in range(n2):
            eng.begin_substep('v', True)
            for G in groups:
                eng.discriminator_step(G)
        lv = eng.loss_v().clone()
        tick = lap('substeps', tick)
        points_after, nxt_domain, nxt_points = ahead.result()
        ahead = pool.submit(draw_ahead, nxt_domain, k + 1 == last) if k < last else None
        tick = lap('sampler_wait', tick)
        L2 = self._l_norm(points_after, domain.V(), as_tensor=True)
        tick = lap('diagnostic', tick)
        prepared = failed = None
        if nxt_points is not None:
            try:
                prepared = self._prepare_groups(nxt_points, nxt_domain, pairs_last=True)
                self._group_cache = prepared
            except Exception as e:        # (e.g. the structure guard on the NEXT sample: this iteration's results are written out first)
                failed = e
        tick = lap('load_next', tick)
        row = torch.cat([torch.stack(lu).reshape(-1).double(), lv.reshape(1).double(), L2.reshape(1)]).tolist()   # the ONE read-back
        tick = lap('read_back', tick)
        ng, keys = len(groups), None
        for i in range(n1):
            vals = row[i * ng:(i + 1) * ng]
            self.last_loss_u = vals[-1]
            self.av_l = 0
            for x_ in vals:
                self.av_l += x_                   # (summed in group order, like the reference's running sum)
            past_losses.append(self.av_l)
            main = self._is_main()            # (several ranks: identical values everywhere, rank 0 writes the files)
            if main:
                past_losses.write('losses_NODE_' + str(d) + '.json')
            if self.av_l < self.best_l:
                if main:
                    keys = keys or self._state_dict_layout()
                    torch.save(self._state_dict_from(snaps[i], keys), 'best_model_weights_NODE.pth')   # u_net as it was after sub-iteration i
                self.best_l = self.av_l
        self.last_loss_v, self._last_L2 = row[n1 * ng], row[n1 * ng + 1]
        times.append(time.time())
        if self._is_main():
            with open('L2_NODE_' + str(d) + '.json', 'w') as fh:
                json.dump([self._last_L2], fh)
            times.write('Time_NODE_' + str(d) + '.json')
        lap('files', tick)
        if failed is not None:
            raise failed
        return prepared, nxt_domain, nxt_points, ahead

    def _iterate(self, report, report_it, show_plt, past_losses, times, pool, draw_ahead):
        """the loop that synchronises after every sub-iteration like the reference's (a `stop` hook, report=True, list domains
        without the sampling thread).  Its side-effect files go through the ordered writer thread (_Saver): same files, same
        order, same final contents, written while the GPU already works on the next sub-iteration."""
        saver = _Saver(self._is_main())
        failing = False
        try:
            return self._iterate_body(report, report_it, show_plt, past_losses, times, pool, draw_ahead, saver)
        except BaseException:
            failing = True
            raise
        finally:
            saver.close(raising=not failing)

    @staticmethod
    def _pinned(loader):
        """page-locked staging of a compact sample (same values): its uploads are asynchronous -- a copy from pageable memory
        makes the host wait for everything queued on the device"""
        return loader.pin() if hasattr(loader, 'pin') else loader

    def _host_state_dict(self, keys):
        """u_net.state_dict() as host tensors, from one small read-back of the parameter blob"""
        return self._state_dict_from(self.engine.theta.data.cpu(), keys)

    def _iterate_body(self, report, report_it, show_plt, past_losses, times, pool, draw_ahead, saver):
        d = self.setup['dim']
        eng = self.engine
        nxt_domain = nxt_points = ahead = prepared = None
        last = self.iterations - 1
        keys = None
        if hasattr(pool, 'first') and self.iterations > 0:        # (the sampling process owns the generators while train() runs)
            nxt_domain, nxt_points = pool.first()
        with torch.cuda.device(self.device):
            for k in range(self.iterations):
                domain = nxt_domain if nxt_domain is not None else self._new_domain()
                points = nxt_points if nxt_points is not None else self._pinned(self._loader(domain))
                nxt_domain = nxt_points = None
                if ahead is None and pool is not None:     # (later ones are submitted the moment the previous result is taken: the
                    ahead = pool.submit(draw_ahead, domain, k == last)   # draws of a ball-domain sample take as long as its sub-steps)
                # (the reference also evaluates L_norm here, src/training.py:123, and overwrites the value unread at :167;
                #  the call draws no random numbers and writes nothing, so it is not repeated)
                if prepared is not None:                   # (list domains: loaded behind the previous iteration's sub-steps)
                    groups, prepared = prepared, None
                else:
                    groups = self._prepare_groups(points, domain)
                self._group_cache = groups
                several = len(groups) > 1
                for G in groups:
                    G.persistent = not several        # list domains: group shapes change every sample -> no graph capture
                if several and ahead is not None and self.defer_list_readback:
                    prepared, nxt_domain, nxt_points, ahead = self._list_iteration_deferred(groups, domain, ahead, pool, draw_ahead, k, last,
                                                                                            past_losses, times)
                    if report and k % report_it == 0:
                        print('iteration: ' + str(k), 'Loss u: ' + str(self.last_loss_u), 'Loss v: ' + str(self.last_loss_v))
                        if self.func_u_sol is not None:
                            print('L^2 norm error: ' + str(self._last_L2))
                            from utils.auxillary_funcs import proj
                            proj(self.u_net, self.setup, k, self.device, axes=[0, 1], resolution=200, colours=20, save=True,
                                 show=show_plt, func_u_sol=self.func_u_sol)
                    continue
                for _ in range(self.n1):
                    self.av_l = 0
                    eng.begin_substep('u', several)
                    lu = []
                    for G in groups:
                        eng.generator_step(G)
                        lu.append(eng.loss_u().clone() if several else eng.loss_u())
                    lu = torch.stack(lu).tolist() if several else [lu[0].item()]     # ONE read-back per sub-iteration
                    self.last_loss_u = lu[-1]
                    self.av_l = 0
                    for x_ in lu:
                        self.av_l += x_                   # (summed in group order, like the reference's running sum)
                    past_losses.append(self.av_l)
                    saver.submit([(_write_text, ('losses_NODE_' + str(d) + '.json', past_losses.text()))])
                    if self.stop is not None and self._stop_agreed(points, domain, None if several else groups[0]):
                        if self._is_main():
                            keys = keys or self._state_dict_layout()
                            saver.submit([(torch.save, (self._host_state_dict(keys), self.path + 'best_model_weights_NODE.pth'))])
                        saver.close()                     # (everything is on disk before the process may end)
                        print('Stopping Criterion Reached')
                        if self.exit_on_stop:
                            exit()
                        return past_losses
                    if self.av_l < self.best_l:
                        if self._is_main():
                            keys = keys or self._state_dict_layout()
                            saver.submit([(torch.save, (self._host_state_dict(keys), 'best_model_weights_NODE.pth'))])
                        self.best_l = self.av_l
                # (no sampling thread -- a `stop` hook may draw random numbers of its own between the sub-iterations --: the host
                #  draws of this point of the loop, the diagnostic's sample and then the next iteration's domain and sample, are
                #  made in the reference's order but BEFORE the read-backs that would wait for the GPU: behind the queued
                #  discriminator sub-step and behind the queued diagnostic, 0.6 ms per outer iteration at the headline size)
                for i_ in range(self.n2):
                    eng.begin_substep('v', several)
                    for G in groups:
                        eng.discriminator_step(G)
                    if ahead is None and i_ == self.n2 - 1:
                        points = self._pinned(self._loader(domain, interior_only=True))
                    self.last_loss_v = eng.loss_v().item()
                if ahead is not None:
                    points, nxt_domain, nxt_points = ahead.result()
                    ahead = pool.submit(draw_ahead, nxt_domain, k + 1 == last) if k < last else None
                elif self.n2 == 0:
                    points = self._pinned(self._loader(domain, interior_only=True))
                if self.capture_refill and not several:
                    L2 = self._l_norm_replayed(groups[0], points, domain)
                    if pool is None and k < last:
                        nxt_domain = self._new_domain()
                        nxt_points = self._pinned(self._loader(nxt_domain))
                    L2 = L2.item()
                else:
                    L2 = self._l_norm(points, domain.V())
                times.append(time.time())
                saver.submit([(_write_text, ('L2_NODE_' + str(d) + '.json', json.dumps([L2]))),
                              (_write_text, ('Time_NODE_' + str(d) + '.json', times.text()))])
                if report and k % report_it == 0 and self._is_main():
                    print('iteration: ' + str(k), 'Loss u: ' + str(self.last_loss_u), 'Loss v: ' + str(self.last_loss_v))
                    if self.func_u_sol is not None:
                        print('L^2 norm error: ' + str(L2))
                        from utils.auxillary_funcs import proj
                        proj(self.u_net, self.setup, k, self.device, axes=[0, 1], resolution=200, colours=20, save=True,
                             show=show_plt, func_u_sol=self.func_u_sol)
        return past_losses

    # --------------------------------------------------------------------------------------------------------------
    def save_checkpoint(self, path):
        """Everything needed to resume: both networks (reference key layout), both Adam states, RNG positions.
        (The reference only ever saves u_net's state_dict, src/training.py:143,148.)"""
        import numpy as np
        torch.save({'u_net': self.u_net.state_dict(), 'v_net': self.v_net.state_dict(),
                    'optimizer_u': self.optimizer_u.state_dict(), 'optimizer_v': self.optimizer_v.state_dict(),
                    'best_l': self.best_l, 'torch_rng': torch.get_rng_state(), 'numpy_rng': np.random.get_state(),
                    'params': dict(self.params)}, path)

    def load_checkpoint(self, path):
        import numpy as np
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.u_net.load_state_dict(ck['u_net'])          # copies into the parameter views, i.e. into the blobs
        self.v_net.load_state_dict(ck['v_net'])
        self.u_net.module.blob.check_alias()
        self.v_net.module.blob.check_alias()
        self.optimizer_u.load_state_dict(ck['optimizer_u'])
        self.optimizer_v.load_state_dict(ck['optimizer_v'])
        self.best_l = ck['best_l']
        torch.set_rng_state(ck['torch_rng'].cpu())
        np.random.set_state(ck['numpy_rng'])
        self.engine.invalidate_test_net()                # cached / prefetched test-network outputs are stale now

    def _stop_agreed(self, points, domain, G=None):
        """the user's stop hook (src/training.py:142).  With several ranks every rank calls it (it may run u_net, which
        every rank can do), but the ranks must leave the loop TOGETHER -- one that returned while the others entered the next
        captured all-reduce would deadlock them -- so rank 0's verdict is the one all of them follow (with rank-local
        sampling the hook sees different paths on every rank and the verdicts can differ).

        What the hook is handed as `interior_points`: the reference's loader keeps its sample on the training device
        (src/dataset.py:300-310), so the hook gets device tensors there too.  For a compact cube sample that is the path tensor
        built on the device from the [N, d] points (328 KB uploaded instead of 11 MB of paths at the headline size), ONE object
        per sample -- so that the exact solution is evaluated once per sample (utils._exact) -- and while the hook runs
        `u_net(that tensor)` is the stepper's forward kernel on the loaded group's own buffers (nets.XNODE._served)."""
        X, mod = None, self.u_net.module
        comp = points.compact() if hasattr(points, 'compact') else None
        if (G is not None and comp is not None and not comp[0].is_cuda and not self.tabulate_on_host and self.world is None
                and G.L > 1 and G.tpp is None):
            held = self.__dict__.get('_hook_sample')
            if held is None or held[0] is not points:
                st = G.__dict__.get('_refill_in')
                fresh = st is not None and getattr(G, '_refill_points', None) is points
                td, du = (st[0], st[1]) if fresh else (self._up(comp[0]), self._up(comp[1]))
                held = self._hook_sample = (points, sampling._paths(td, du))
            X = held[1]
            mod._served = (X, lambda: self.engine.predict_group(G))
        try:
            verdict = bool(self.stop(self, X if X is not None else points.interioru, domain))
        finally:
            mod._served = None
        if self.world is None:
            return verdict
        flag = torch.tensor([1.0 if (verdict and self.world.rank == 0) else 0.0], dtype=torch.float64, device=self.device)
        self.world.all_reduce(flag)
        return bool(flag.item() > 0.5)

    def _is_main(self):
        return self.world is None or self.world.rank == 0
