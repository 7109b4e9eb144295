"""The fused XNODE-WAN training step: generator and discriminator sub-steps as explicit sequences of HIP kernels.

Replaces the bodies of the two sub-step loops of NODE_WAN_solver.train (src/training.py:125-138,151-162 of the
reference): forward of both nets, func_eval, loss.u / loss.v with their three autograd backward passes, and
optimizer.step().  No autograd tape is built; every gradient is an explicit kernel (xw_ode_bwd, xw_disc_bwd).

Semantics are the reference's ON A GPU, stated once here and implemented literally below:
  Q1  the helper backwards of loss.I also deposit d(sum u)/dtheta resp. d(sum phi)/dphi in the parameter gradients
      -> `pollution = 1` in the cotangent kernels (set Engine.pollution = 0.0 for the textbook gradient);
  Q2  nabla u, nabla phi enter I as constants; the u-factor of the d(phi)/dt term carries no gradient;
  Q3  nabla_x u is the time-summed input gradient G_n = d(sum_l u[n,l])/dx_n (incl. the path through h(x)), seen at
      time index 0 only -> the a_ij contraction is per path (s3x);
  Q4  v is evaluated on its own interior sample XV;
  Q5  no stale .grad carry-over between sub-steps (that is a CPU-only artefact of the reference).

Data layout in HBM (per group of N equal-length paths, L sample times, d dimensions):
  xT, xvT, xbT  float64 [d, N]   transposed coordinates of the u-, v- and boundary samples (the [N, L, d+1] path tensor
                                 of the reference is never materialised on the device: on vertical paths it is x (x) t)
  t             float64 [L]      shared time grid
  u, v, vt, f, ubar, vbar ...    float64 [L, N]   time-major point arrays (coalesced for one-lane-per-path kernels)
  Y             float64 [L, H, N]  hidden-state checkpoints of the stepper (written by the forward, read by the sweeps)
  act, act_b    float64 [L-1, rows, N]   stage activations of every step (forward -> sweeps: no field re-evaluation)
  vact          float64 [(q+1) W, N L]   layer inputs of the test network (its forward -> its backward)
  slabs         float64 [n_slab, P]  per-wave partial parameter gradients, summed inside the Adam kernel
"""
import contextlib
import math
import os

import torch

from . import kernels as KN
from ._lib import XnwanError
from .options import EngineOptions
from .sampling import HIP_HOST_LOCK, _PIN_POOL, Hypercube, _paths


def ctypes_addr(fn):
    """address of a ctypes callback object (kept alive by the caller)"""
    import ctypes
    return ctypes.cast(fn, ctypes.c_void_p).value

# Every captured sub-step graph of the process, kept alive until it exits.  On this stack (ROCm 7.2 runtime inside the
# PyTorch 2.10 wheel) destroying the executable of a multi-branch graph -- which is what Python's garbage collector does to
# the graphs of a solver that went out of scope -- leaves the runtime's per-graph stream bookkeeping in a state in which a
# LATER launch of another, live graph dereferences a dead stream: SIGSEGV in hip::Graph::UpdateStreams under
# hipGraphLaunch (rocgdb backtrace, round 3; it took a particular sequence of 27 tests to line the collector up with a
# replay).  Collecting before a new engine captures does not help; not destroying does.  A graph is a few dozen kernel
# nodes and holds no sample buffers (those belong to the group), so the cost of keeping it is small.
_KEPT_GRAPHS = []

_NOSTREAM = contextlib.nullcontext()      # Engine._side without side streams

# One private memory pool for the captured graphs that ALLOCATE while they are recorded (the training loop's refill of its
# group, its diagnostic: path tensors and the temporaries of the user's callables, ~100 MB at the headline size).  Their
# results are copied into buffers of the group inside the graph, nothing in the pool is read after a replay has ended, and
# replays are issued on one stream: the graphs of every group and solver of the process can reuse the same memory -- kept
# graphs (above) then cost their nodes, not a pool each.
_SCRATCH_POOL = []


# The side streams and the capture stream are per DEVICE, shared by every engine of the process (engines are driven from one
# host thread and never run concurrently): the allocator only reuses a block on the stream it was allocated on, so with
# streams per engine the scratch pool above grew by what one refill allocates (54 MB at the headline size) per solver.
_DEVICE_STREAMS = {}


def _device_streams(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _DEVICE_STREAMS:
        _DEVICE_STREAMS[key] = ([torch.cuda.Stream(device=device) for _ in range(4)], torch.cuda.Stream(device=device))
    return _DEVICE_STREAMS[key]


def _scratch_pool(which=0):
    """which: graphs that may be replayed CONCURRENTLY (the diagnostic of one outer iteration beside the refill for the next one,
    solver._iterate_pipelined) must not share scratch memory -- pool 0: refills, pool 1: diagnostics"""
    while len(_SCRATCH_POOL) <= which:
        _SCRATCH_POOL.append(torch.cuda.graph_pool_handle())
    return _SCRATCH_POOL[which]

F32, F64 = torch.float32, torch.float64


def _d64(x, dev):
    """x on `dev` as float64; the tensor itself when it already is (two no-op .to() calls cost ~8 us of host time, and a
    list-domain sample makes ~400 of them)"""
    if x.dtype == F64 and x.device == dev:
        return x
    return x.to(dev).to(F64)


def _to_LN(x, dev):
    """[N, L] (any float dtype, any device) -> contiguous float64 [L, N] on dev"""
    return _d64(x.detach(), dev).t().contiguous()


class Structure:
    """What the PDE coefficient callables look like, found by probing them once on a few random points."""

    def __init__(self, funcs, d, lo=-1.0, hi=1.0):
        g = torch.Generator().manual_seed(20221111)
        Xp = torch.rand(6, 3, d + 1, generator=g) * (hi - lo) + lo
        Xp[:, :, 0] = torch.rand(6, 3, generator=g)
        ident = True
        for i in range(d):
            for j in range(d):
                a = funcs['a'](Xp, i, j)
                if not bool(torch.all(a == (1.0 if i == j else 0.0))):
                    ident = False
                    break
            if not ident:
                break
        self.a_identity = ident
        self.b_zero = all(bool(torch.all(funcs['b'](Xp, i) == 0)) for i in range(d))
        u1 = torch.randn(6, 3, 1, generator=g, dtype=F64)
        u2 = torch.randn(6, 3, 1, generator=g, dtype=F64)
        c1, c2 = funcs['c'](Xp, u1), funcs['c'](Xp.flip(0), u2)
        k = (c1 / u1).reshape(-1)
        self.c_kappa = None
        if torch.is_tensor(c1) and c1.shape == u1.shape and bool(torch.all(k == k[0])) and bool(torch.all(c2 == k[0] * u2)):
            self.c_kappa = float(k[0])

    def describe(self):
        return 'a=%s b=%s c=%s' % ('identity' if self.a_identity else 'general', 'zero' if self.b_zero else 'general',
                                   ('%g*u' % self.c_kappa) if self.c_kappa is not None else 'general')


class Group:
    """Device-resident data of one group of equal-length paths + every buffer its sub-steps need.  Buffers are allocated
    once and refilled in place by Engine.load_group(..., into=G), so captured HIP graphs stay valid across resampling."""
    SAMPLE_FIELDS = ('t', 'tb', 'tpp', 'tpp0', 'xvT_pts', 'xT', 'xvT', 'xbT', 'start', 'ghT', 'h', 'href', 'f', 'w', 'wt', 'w0', 'gwx0T',
                     'start_b', 'g', 'X', 'A0', 'B0')

    def signature(self):
        return tuple((k, tuple(getattr(self, k).shape)) for k in self.SAMPLE_FIELDS if getattr(self, k, None) is not None)

    # The work buffers are regions of ONE allocation (`_arena`), described by `_lazy` = {name: (offset, shape)}; the tensor
    # view of a region is made when somebody reads the attribute.  The sub-step runner only needs addresses (ptr): the groups
    # of a list domain are rebuilt with every sample, 19 groups x 22 buffers, and a torch.empty or a view costs the host 4 us each.
    def __getattr__(self, name):          # (only reached when the attribute is not set)
        lazy = self.__dict__.get('_lazy')
        if lazy is not None and name in lazy:
            off, shape = lazy[name]
            t = self.__dict__['_arena'][off:off + math.prod(shape)].view(shape)
            self.__dict__[name] = t
            return t
        raise AttributeError(name)

    def ptr(self, name):
        """device address of a buffer (0: the group has none)"""
        t = self.__dict__.get(name)
        if t is not None:
            return t.data_ptr()
        lazy = self.__dict__.get('_lazy')
        if lazy is not None and name in lazy:
            return self.__dict__['_arena'].data_ptr() + 8 * lazy[name][0]
        return 0


class Engine:
    def __init__(self, config, setup, u_mod, v_mod, funcs, device, world=None, structure=None, options=None):
        self.config, self.setup, self.u, self.v, self.funcs, self.dev, self.world = config, setup, u_mod, v_mod, funcs, device, world
        # every switch comes from ONE object (options.EngineOptions; the XW_* environment is read once, in from_env()); the
        # attributes below are the working copies -- tests and tools set them directly
        opt = self.options = options if options is not None else EngineOptions.from_env()
        self.d = setup['dim']
        self.method = KN.method_id(config['solver'])
        # config['adjoint'] (src/model.py:103): the sweeps integrate torchdiffeq's continuous adjoint instead of reversing
        # the steps taken (include/xnwan.h, xw_ode_bwd mode bit 3); nabla_x u then only flows through the start value
        self.adjoint = bool(config.get('adjoint', False))
        self.alpha = float(config['alpha'])
        self.pollution = 1.0
        self.verify_structure = opt.verify_structure    # (_check_structure)
        self.packed_load = opt.packed_load     # list domains: load_groups_packed
        self.refill_variants = 8        # captured variants of a group's refill / diagnostic graph before further ones run eagerly
        self.verify_every = 16          # ~3 d + 1 callable evaluations per check: ~1 ms at d = 20, a fifth of an outer iteration
        sp = setup.get('shape_param', [-1, 1])
        lo, hi = (sp[0], sp[1]) if isinstance(sp, (list, tuple)) else (-sp, sp)
        self.structure = structure if structure is not None else Structure(funcs, self.d, lo, hi)
        if u_mod.blob is None or v_mod.blob is None:
            raise XnwanError('bind() the networks to the device before building the engine')
        # widths of the kernel instantiations the two networks run in (>= the configured widths: nets.Blob embeds a narrower
        # network exactly, zero-padded, in the next larger instantiation)
        (self.H, self.K), self.m = u_mod.kdims, config['u_layers']
        self.W, self.q = v_mod.kwidth, config['v_layers']
        # widths beyond the MFMA containers run on the generic path (csrc/xw_generic.hip): correct, deterministic, and two to three
        # orders of magnitude slower -- said once, loudly
        self.generic = (KN.ode_generic(self.H, self.K, self.m), KN.disc_generic(self.W))
        if any(self.generic):
            import warnings
            which = ' and '.join(n_ for n_, g_ in zip(('u_theta (u_hidden_dim %d, u_hidden_hidden_dim %d, u_layers %d)' % (self.H, self.K, self.m),
                                                       'v_phi (v_hidden_dim %d)' % self.W), self.generic) if g_)
            warnings.warn('%s is outside the MFMA kernel instantiations %s (u_layers <= 10) / %s: running on the generic vector-ALU path, expect a step '
                          'rate lower by two to three orders of magnitude' % (which, KN.ODE_WIDTHS, KN.DISC_WIDTHS), RuntimeWarning, stacklevel=3)
        # the test network's input layer: spatial columns once per path (xw_disc_xproj) -- the MFMA widths, paths over a shared grid
        # XW_XPROJ_MIN_D: from which d on.  In the sub-step cycle the split form wins from d ~ 45 on and not below, at 131072 points as at
        # a million (profiles/r05_xproj.txt: headline d = 20 0.488 against 0.473 ms per sub-step -- the small launch is one more dependent
        # node on the critical chain --, d = 20 at 16384 x 64 3.41 against 3.38; d = 50 -1.3 %, BASELINE configs[2] -2.7 %, [3] -6.5 %).
        self.xproj_min_d = 1 << 30 if self.generic[1] else int(opt.xproj_min_d)
        if self.W > 64 and not self.generic[1]:
            self.xproj_min_d = 0        # (the 128-wide container: 131 KB of Vh fragments leave no LDS for input-layer fragments -- always the table)
        if self.generic[0] and self.adjoint:
            raise XnwanError('adjoint=True (the continuous adjoint) exists for the MFMA stepper instantiations %s only; u_hidden_dim = %d, '
                             'u_hidden_hidden_dim = %d run on the generic path, which reverses the steps taken (adjoint=False)'
                             % (KN.ODE_WIDTHS, self.H, self.K))
        self.theta, self.phi = u_mod.blob, v_mod.blob
        self.Pu, self.Pv = self.theta.data.numel(), self.phi.data.numel()
        z = lambda n: torch.zeros(n, dtype=F64, device=device)  # noqa: E731
        self.adam_u = dict(m=z(self.Pu), v=z(self.Pu), step=torch.zeros(1, dtype=torch.int64, device=device),
                           lag=torch.zeros(1, dtype=torch.int64, device=device))
        # Reference behaviour on the single-slice groups of the list domains (SURVEY A.4 / DESIGN 8 "Q8"), both on by default:
        #  * NeuralODE.forward returns [N,1] instead of [N,1,1] on a single-slice group at T0 (src/model.py:89-91) and the loss
        #    then broadcasts [N] against [N,1] into [N,N] tables over all PAIRS of paths (src/loss.py:65,70,79,84): reproduced
        #    in factorised O(N) form (load_group / xw_weak_partials pairwise).  XW_ELEMENTWISE_SINGLE_SLICE=1: the
        #    elementwise expressions instead (what the formulas mean; NOT what the reference computes).
        #  * such a group never integrates the ODE, so the field's parameters get no gradient; after zero_grad() (None on
        #    torch >= 2.0) Adam SKIPS them -- no moment decay, no step count -- until a group of the sub-iteration has taken an
        #    ODE step: the field range of the blob keeps its own step count (xw_adam lag / skip).
        self.pairwise_single_slice = opt.pairwise_single_slice
        self.adam_skips_untouched = opt.adam_skips_untouched
        self.eager_checked = 10 ** 12 if opt.always_check else 256
        self.field_range = (self.theta.slots[6][0], self.theta.slots[-2][0])      # Win .. Wo.b (nets._u_slots order)
        self._field_touched = False
        self.adam_v = dict(m=z(self.Pv), v=z(self.Pv), step=torch.zeros(1, dtype=torch.int64, device=device))
        self.grad_u, self.grad_v = z(self.Pu), z(self.Pv)   # the gradient Adam saw in the last sub-step
        # generator exchange buffer [sum of A slabs | sum of B slabs | scal]: ONE all-reduce per generator sub-step
        self.pack_u = z(2 * self.Pu + 16)
        self.scal = self.pack_u[2 * self.Pu:]
        # gradient carried from the previous groups of the same sub-iteration (list domains: the reference calls zero_grad()
        # once per sub-iteration but optimizer.step() after every group, src/training.py:127-138); None = off (one group)
        self.accum_u = self.accum_v = None
        self.use_streams = opt.use_streams   # independent kernel chains on side streams
        self.use_graphs = opt.use_graphs     # capture each sub-step into a HIP graph and replay it
        # opt-in: v, dv/dt, nabla_x v(t_0) of a group are reused while phi and the sample are unchanged (exact: the
        # reference recomputes identical values in every sub-step of an outer iteration).  Off by default.
        self.reuse_test_net = opt.reuse_test_net
        # both forwards store their layer inputs for their backwards (include/xnwan.h: XwOdeFwdJob.act, xw_disc_fwd act)
        self.keep_activations = opt.keep_activations
        # The test network's launch is persistent (grid-stride over point tiles) and at 2 blocks per CU it owns every SIMD's
        # register file: the stepper's waves, launched next to it, then wait until it drains.  Capping it below the
        # resident slots leaves SIMDs to the stepper chains.  With the kernel's ticket queues (tiles go to whichever wave is
        # free) the generator sub-step is flat between 9/16 and 11/16 of the slots (0.518 - 0.521 ms; 0.527 at 3/4, 0.533 at
        # 1/2); the static split needed exactly 3/4 (0.5225 ms, 0.58 either side).  Round 3 (leaner test network and stepper
        # forward): 10/16 -- cycle 1.546 / 1.541 / 1.554 / 1.557 ms at 9/16, 10/16, 11/16, 12/16.
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        # Round 4: with the generator's sweeps A + boundary at lowered wave priority (prio_drop below) the test network keeps more of
        # the chip: generator sub-step 0.492 / 0.484 / 0.480 / 0.475 ms at 9, 10, 11, 12 sixteenths of the slots -- and 0.545 at 13 (a
        # cliff: the stepper's forward pass no longer finds SIMDs); 12/16.
        # (the 96- / 128-wide containers run one block per CU, the cap is clipped to all of them: 3/4 of the CUs measured the same with
        #  the wide stepper beside it and 6 % slower with the narrow one -- profiles/r06_width_step_rate.txt)
        self.v_blocks = int(opt.v_blocks) or (12 * 2 * cus) // 16
        # (discriminator sub-step: only the stepper forward and the x-only sweep run beside it, 33 us of SIMD time: 7/8 of the
        #  slots -- 0.615 ms against 0.638 at 3/4, 0.681 at 15/16, 0.735 at all of them; round 3: 13/16 and 14/16 equal
        #  (cycle 1.537 / 1.538 ms), 15/16 1.626; with the record stored through global instead of flat instructions the
        #  forward is 8 % shorter and the stepper's chain is what the sub-step waits for: 12/16 -- discriminator sub-step
        #  0.586 / 0.562 / 0.566 / 0.569 / 0.654 ms at 11..15 sixteenths, tools/sweep_caps.py)
        #  round 4: 0.583 / 0.566 / 0.563 / 0.566 ms at 11..14 sixteenths: 13/16)
        self.v_blocks_disc = int(opt.v_blocks_disc) or (13 * 2 * cus) // 16
        # Narrow tiles (csrc/xw_ode_n4.h, xw_ode_bwd mode bit 4): a 16-path tile of a sweep as four waves of 4 paths instead of one
        # (+ a partner) -- four times the instruction streams, each a shorter chain, at ~1.8 x the matrix-pipe time per path.
        # Used where a sweep runs with SIMDs to spare: sweep B of the generator sub-step (alone on the chip behind the test
        # network), as long as its waves still find a SIMD each.  XW_NARROW: 0 off, 1 auto (default), 2 wherever possible.
        self.narrow = str(opt.narrow)
        # wave-priority drops of the stepper launches that are not on a sub-step's critical path (include/xnwan.h: xw_ode_bwd mode
        # bits 5..6, XwOdeFwdJob.prio_drop): A = the generator's sweeps A + boundary, X = the discriminator's x-only sweep,
        # F = the discriminator's forward pass, G = the generator's.  A = 3 puts sweeps A + boundary at the test network's own
        # priority (0): 0.4764 against 0.4838 ms per sub-step at 2 (tools/cap_sweep.sh, three alternating runs each; found at the end
        # of round 4: the first sweep stopped at 2); G = 1..2 and X, F = 1 are within the noise of that, G = 3 loses 7 %.
        # At larger d the test network's launch is longer relative to the stepper's chains (its input layer and the fused
        # nabla_x v grow with d, the stepper's x-projection is hoisted) and sweeps that never get ahead of it end AFTER sweep B:
        # tools/ab_cfg_prio.sh, priority 1 against 0 -- d = 20: -1.4 % / -1.1 % at 4096 / 8192 paths, equal at 2048; d = 50: equal
        # at 2048 x 64, +0.8 % at 16384 x 64; d = 100: +1.2 % at 8192, +1.3 % at 65536.  Default: 3 up to d = 32, 2 above.
        # The sub-step graphs' four chains are laid out for the HIP runtime's default of FOUR hardware queues: with 5, 6 or 8
        # (GPU_MAX_HW_QUEUES) the headline sub-step takes 0.69 instead of 0.47 ms (tools/hwq.sh; 2 and 3 are as good as 4).
        if opt.hw_queues > 4:
            import warnings
            warnings.warn('GPU_MAX_HW_QUEUES=%s: the sub-step schedule is tuned for the runtime default of 4 hardware queues and '
                          'measures ~45 %% slower with more' % opt.hw_queues, RuntimeWarning, stacklevel=2)
        self.early_slab_sum = opt.early_slab_sum
        # groups of at most this many 16-path tiles (interior + boundary) take the compact schedule of _gen_front_compact (0: never).
        # tools/shard_streams.sh, ms per sub-step wide / compact: 256 paths (+ 256 boundary paths) 0.2489 / 0.2183, 512 0.2609 / 0.2244,
        # 1024 0.2829 / 0.2487, 1536 0.3190 / 0.3108, 2048 0.3490 / 0.3432, 2560 0.3931 / 0.3791, 3072 0.4173 / 0.4272, 4096 0.4698 / 0.4919
        self.compact_tiles = int(opt.compact_tiles)
        self.prio_drop = {'A': int(opt.prio_drop_A) if opt.prio_drop_A is not None else (3 if self.d <= 32 else 2),
                          'X': int(opt.prio_drop_X), 'F': int(opt.prio_drop_F), 'G': int(opt.prio_drop_G)}
        self.use_runner = opt.use_runner      # one C call per eager group sub-step (xw_substep_*)
        # Measured (profiles/r04_shard_sweep.md): forward and the sweep without weight gradients gain on shards up to ~2048
        # paths (0.302 -> 0.272 ms per sub-step at 512 paths, 0.332 -> 0.294 at 1024, 0.375 -> 0.367 at 2048); the narrow sweep
        # WITH weight gradients only ties the two-wave duo sweep (88 against 83 us alone) and is left to XW_NARROW_SET=fxp; at
        # the headline size everything narrow LOSES (0.502 -> 0.586 ms): those phases are bound by the sum of SIMD time.
        self.narrow_set = str(opt.narrow_set)
        self.narrow_tiles = {'f': 192, 'x': 128, 'p': 64}    # largest launch (16-path tiles, all its jobs) that still gains
        if opt.narrow_tiles:                                 # (measurements: "f:x:p")
            self.narrow_tiles = dict(zip('fxp', (int(v_) for v_ in str(opt.narrow_tiles).split(':'))))
        # ... and for a launch that has the chip to itself (tools/kernel_times.py at 256 / 512 tiles, profiles/r06_kernel_times.txt)
        self.narrow_tiles_alone = dict(zip('fxp', (int(v_) for v_ in str(opt.narrow_tiles_alone).split(':'))))
        for k_ in 'fxp':
            self.narrow_tiles_alone[k_] = max(self.narrow_tiles_alone[k_], self.narrow_tiles[k_])
        self.simds = 4 * cus
        self._phi_version = 0
        self.streams, self._cap = _device_streams(device)
        # several GPUs on RCCL: the exchanges are device-side calls on the current stream (dist.World.capturable), so a
        # sub-step and its exchange(s) are captured into ONE HIP graph instead of graph / host call / graph
        self.capture_exchange = (world is not None and getattr(world, 'capturable', False)
                                 and opt.capture_exchange)
        if self.capture_exchange:
            world.all_reduce(self.scal)           # (zeros) first use outside any capture: RCCL sets up its channels here

    # ------------------------------------------------------------------------------------------------------------
    # coefficient tables (src/training.py:32-41).  Only time index 0 can contribute (Q3), so that slice is all that is
    # tabulated: [d,d,N] instead of the reference's [d,d,N,L].
    # ------------------------------------------------------------------------------------------------------------
    def _tabulate_a(self, X1):
        """a_ij(t_0, x_n) on the sample X1 [N,1,d+1], returned as (table, amode) -- the form is stated explicitly
        (xw_weak_contract_general's amode), never guessed from the shape: [d,d] and [d,N] coincide when N == d.  ONE batched call of the user's callable with index tensors
        i[d,1,1,1], j[1,d,1,1] when it is written with tensor operations (checked against scalar calls on three index
        pairs); otherwise the reference's d^2 scalar calls.  The table is then stored in the cheapest EXACT form: one
        [d,d] matrix if it does not vary over the sample, its diagonal [d,N] if every off-diagonal entry is exactly zero."""
        d, dev, fa = self.d, self.dev, self.funcs['a']
        N = X1.shape[0]
        A = None
        try:
            ii = torch.arange(d, device=X1.device).view(d, 1, 1, 1)
            jj = torch.arange(d, device=X1.device).view(1, d, 1, 1)
            out = fa(X1.unsqueeze(0).unsqueeze(0), ii, jj)
            if torch.is_tensor(out) and tuple(out.shape) == (d, d, N, 1):
                A = out[..., 0].to(dev).to(F32).to(F64)       # (the reference's table is float32: src/training.py:32)
                g = torch.Generator().manual_seed(d * 7919 + N)
                for _ in range(3):
                    i, j = (int(k) for k in torch.randint(0, d, (2,), generator=g))
                    if not torch.equal(A[i, j], fa(X1, i, j).to(dev).to(F32).to(F64)[:, 0]):
                        A = None
                        break
        except Exception:               # the callable branches on (i, j) in Python, indexes with them, ...: scalar calls
            A = None
        if A is None:
            A = torch.stack([torch.stack([fa(X1, i, j).to(dev).to(F32).to(F64)[:, 0] for j in range(d)], 0) for i in range(d)], 0)
        off = A.clone()
        off.diagonal(dim1=0, dim2=1).zero_()
        if bool(torch.all(A == A[:, :, :1])):
            return A[:, :, 0].contiguous(), 1                          # [d, d]   one matrix for all points
        if bool(torch.all(off == 0)):
            return A.diagonal(dim1=0, dim2=1).t().contiguous(), 2      # [d, N]   diagonal
        return A.contiguous(), 3                                       # [d, d, N] full table

    def _tabulate_b(self, X1):
        d, dev, fb = self.d, self.dev, self.funcs['b']
        N = X1.shape[0]
        try:
            out = fb(X1.unsqueeze(0), torch.arange(d, device=X1.device).view(d, 1, 1))
            if torch.is_tensor(out) and tuple(out.shape) == (d, N, 1):
                B = out[..., 0].to(dev).to(F32).to(F64)       # (float32 table: src/training.py:37)
                if all(torch.equal(B[i], fb(X1, i).to(dev).to(F32).to(F64)[:, 0]) for i in (0, d // 2, d - 1)):
                    return B.contiguous()
        except Exception:
            pass
        return torch.stack([fb(X1, i).to(dev).to(F32).to(F64)[:, 0] for i in range(d)], 0).contiguous()

    def _check_structure(self, X, version):
        """The fused fast paths (a = identity, b = 0, c = kappa u) were chosen from a probe on random points
        (Structure).  Guard them on the ACTUAL sample, every time one is loaded: the whole diagonal of a, one rotating
        off-diagonal per row (all pairs are visited over d samples), every b_i, and c against kappa u -- a coefficient
        that only deviates in part of the domain raises here instead of silently training the wrong PDE."""
        st, d = self.structure, self.d
        X1 = X[:, :1, :]
        bad = []                                   # (device-side flags, ONE host sync for the whole check)
        if st.a_identity:
            for i in range(d):
                j = (i + 1 + version % max(d - 1, 1)) % d
                bad.append(('func_a[%d,%d] == 1' % (i, i), torch.any(self.funcs['a'](X1, i, i) != 1)))
                if d > 1:
                    bad.append(('func_a[%d,%d] == 0' % (i, j), torch.any(self.funcs['a'](X1, i, j) != 0)))
        if st.b_zero:
            for i in range(d):
                bad.append(('func_b[%d] == 0' % i, torch.any(self.funcs['b'](X1, i) != 0)))
        if st.c_kappa is not None:
            key = (tuple(X.shape[:2]), str(X.device))
            if getattr(self, '_probe_u', (None, None))[0] != key:
                g = torch.Generator().manual_seed(1)
                self._probe_u = (key, torch.randn(X.shape[0], X.shape[1], 1, generator=g, dtype=F64).to(X.device))
            up = self._probe_u[1]
            bad.append(('func_c(X, u) == %g u' % st.c_kappa, torch.any(self.funcs['c'](X, up) != st.c_kappa * up)))
        # (results on the host cost nothing to read; results on the device are read back together: ONE sync)
        dev_flags = [b for _, b in bad if b.is_cuda]
        hit = any(bool(b) for _, b in bad if not b.is_cuda) or (bool(torch.stack(dev_flags).any()) if dev_flags else False)
        if hit:
            which = [name for name, b in bad if bool(b)]
            raise XnwanError('the PDE coefficients do not have the structure the probe at construction saw (%s): violated on this '
                             'sample: %s.  Build the solver with an explicit engine.Structure.' % (st.describe(), ', '.join(which[:4])))

    # ------------------------------------------------------------------------------------------------------------
    # per-sample preparation (once per outer iteration; everything here is parameter-independent)
    # ------------------------------------------------------------------------------------------------------------
    def tabulate_sample(self, triples, domain, hints=None, grids=None):
        """List domains (src/dataset.py:48-229: 11-20 groups per sample): evaluate the user's callables h, f, g and the
        domain's weight w (with their input gradients) ONCE on the points of ALL groups instead of group by group, and
        hand every group its slices (load_group(tab=...)).  The callables are PDE data -- functions of the point (t, x) --
        so evaluating them on a concatenation is the same arithmetic per point; that is CHECKED on the first group of the
        first sample (bitwise against the per-group call) and batching is switched off with a warning if it does not hold.
        Per outer iteration this removes ~140 of ~150 callable evaluations, each a dozen tiny kernel launches."""
        if getattr(self, '_batch_tab', True) is False or len(triples) < 2:
            return [None] * len(triples)
        if sum(t_[0].shape[0] for t_ in triples) == 0 or sum(t_[2].shape[0] for t_ in triples) == 0:
            return [None] * len(triples)      # (a rank whose shares of this sample hold no interior / no boundary path at all: group by group)
        d, T0 = self.d, self.setup['T0']
        Xs = [t_[0].detach() for t_ in triples]
        XVs = [t_[1].detach() for t_ in triples]
        BXs = [t_[2].detach() for t_ in triples]
        if hints is not None and all(h is not None for h in hints):      # (the loader read them off its host copies: no read-back)
            first_t = [h['t0'] for h in hints] + [h['tb0'] for h in hints]
        else:
            # ONE host sync for all start times; a share of a sharded group reads the WHOLE group's first paths (`grids`: it may be
            # empty, or start at another path's entry time)
            gr = grids if grids is not None else [None] * len(Xs)
            if any(g_ is None and (x.shape[0] == 0 or b.shape[0] == 0) for g_, x, b in zip(gr, Xs, BXs)):
                raise XnwanError('tabulate_sample: an empty share needs the first times of the whole group (hints or grids)')
            first_t = torch.stack([(g_[0][0] if g_ is not None else x[0, 0, 0]).to(Xs[0].device) for g_, x in zip(gr, Xs)] +
                                  [(g_[1][0] if g_ is not None else b[0, 0, 0]).to(Xs[0].device) for g_, b in zip(gr, BXs)]).tolist()
        at0 = [float(v) == T0 for v in first_t[:len(Xs)]]
        bat0 = [float(v) == T0 for v in first_t[len(Xs):]]
        pts = lambda ts: torch.cat([t_.reshape(-1, 1, d + 1) for t_ in ts], 0)                      # noqa: E731  [P, 1, d+1]
        cuts = lambda ts: [t_.shape[0] * t_.shape[1] for t_ in ts]                                  # noqa: E731
        f_cat = self.funcs['f'](pts(Xs)).detach().reshape(-1)
        g_cat = self.funcs['g'](pts(BXs)).detach().reshape(-1)
        f_all, g_all = f_cat.split(cuts(Xs)), g_cat.split(cuts(BXs))
        XVp = pts(XVs).clone().requires_grad_(True)
        w_all = domain.func_w(XVp)
        gw_all = torch.autograd.grad(w_all.sum(), XVp)[0] if w_all.requires_grad else torch.zeros_like(XVp)
        w_cat, gw_cat = w_all.detach().reshape(-1), gw_all.reshape(-1, d + 1)
        w_all, gw_all = w_cat.split(cuts(XVs)), gw_cat.split(cuts(XVs))
        # start values with their x-gradient: h for groups that start at T0, g for groups that start on the boundary
        # (h is evaluated ONLY on the start points of the groups that start at T0 and g ONLY on those that start on the
        #  boundary -- the combinations the per-group path evaluates: a callable that is not finite, or has no finite
        #  gradient, where it is never asked must not leak a NaN into the other groups through a masked merge)
        def starts(ts, flags):
            P0 = torch.cat([x[:, 0, :] for x in ts], 0).clone().requires_grad_(True)
            n = [x.shape[0] for x in ts]
            # (the groups are contiguous row ranges of P0: the rows of the groups that start at T0 / on the boundary are taken and
            #  put back as slices -- an index tensor would have to be uploaded or found on the device, both of which wait for
            #  everything queued on the stream, i.e. for the previous iteration's sub-steps when this runs beside them)
            rows = P0.split(n)
            pick = lambda want: [r for r, a in zip(rows, flags) if a == want]                 # noqa: E731
            rows_h, rows_g = pick(True), pick(False)
            some = lambda rows: sum(r.shape[0] for r in rows) > 0          # noqa: E731  (a rank's shares may hold none of them)
            hval = self.funcs['h'](torch.cat(rows_h, 0)).reshape(-1) if some(rows_h) else None
            gval = self.funcs['g'](torch.cat(rows_g, 0).unsqueeze(1)).reshape(-1) if some(rows_g) else None
            ref = hval if hval is not None else gval
            none = ref.new_zeros(0)
            it_h = iter(hval.split([r.shape[0] for r in rows_h])) if hval is not None else iter([none] * len(rows_h))
            it_g = iter(gval.to(ref.dtype).split([r.shape[0] for r in rows_g])) if gval is not None else iter([none] * len(rows_g))
            val = torch.cat([next(it_h) if a else next(it_g) for a in flags], 0)
            return P0, n, val
        X0, n0, start = starts(Xs, at0)
        gh = torch.autograd.grad(start.sum(), X0)[0][:, 1:] if start.requires_grad else torch.zeros(X0.shape[0], d, device=X0.device, dtype=X0.dtype)
        # h on every interior start point (the s1 term and the initial penalty read it on all groups, src/loss.py:64,79;
        # for the groups that start at T0 it IS the start value)
        hv = start if all(at0) else self.funcs['h'](X0.detach()).reshape(-1)
        _, nb0, sb = starts(BXs, bat0)
        sb = sb.detach()
        tabs = []
        # (the unsplit tables, for load_groups_packed: one gather launch takes every group's fields out of them)
        self._tab_cat = dict(f=f_cat, g=g_cat, w=w_cat, gw=gw_cat, start=start.detach(), gh=gh, h=hv.detach(), start_b=sb, at0=at0, bat0=bat0)
        start_k, gh_k, hv_k, sb_k = start.detach().split(n0), gh.split(n0), hv.detach().split(n0), sb.split(nb0)   # (once: a split is 20 new tensors)
        for k, (x, xv, bx) in enumerate(zip(Xs, XVs, BXs)):
            N, L = x.shape[0], x.shape[1]
            tabs.append(dict(starts_T0=at0[k], start=start_k[k], gh=gh_k[k], h=hv_k[k],
                             f=f_all[k].view(N, L), w=w_all[k].view(N, L), gw=gw_all[k].view(N, L, d + 1),
                             b_T0=bat0[k], start_b=sb_k[k], g=g_all[k].view(bx.shape[0], bx.shape[1])))
        if not getattr(self, '_batch_tab_checked', False):
            self._batch_tab_checked = True
            # bitwise against the per-group calls: f, h, w on the first group, and g, the start values with their gradient and the
            # boundary start values on the LAST triple (a late group: boundary-type starts on the hourglass)
            # (the first and the last group of which this rank holds interior AND boundary paths: an empty share has nothing to compare)
            full = [k for k in range(len(Xs)) if Xs[k].shape[0] > 0 and BXs[k].shape[0] > 0] or [0]
            k0, kl = full[0], full[-1]
            x, t0 = Xs[k0], tabs[k0]
            xl, bl, tl = Xs[kl], BXs[kl], tabs[kl]
            eq = lambda a, b: torch.equal(a.detach().reshape(-1), b.detach().reshape(-1))      # noqa: E731
            xl0 = xl[:, 0, :].clone().requires_grad_(True)
            s_l = self.funcs['h'](xl0) if at0[kl] else self.funcs['g'](xl0.unsqueeze(1)).reshape(-1)
            g_l = torch.autograd.grad(s_l.sum(), xl0)[0][:, 1:] if s_l.requires_grad else torch.zeros_like(xl0[:, 1:])
            sb_l = self.funcs['h'](bl[:, 0, :]) if bat0[kl] else self.funcs['g'](bl[:, 0, :].unsqueeze(1))
            if not (eq(self.funcs['f'](x), t0['f']) and eq(self.funcs['h'](x[:, 0, :]), t0['h']) and eq(domain.func_w(XVs[k0]), t0['w'])
                    and eq(self.funcs['g'](bl), tl['g']) and eq(s_l, tl['start']) and eq(g_l, tl['gh']) and eq(sb_l, tl['start_b'])):
                import warnings
                warnings.warn('the PDE callables give different values on a concatenation of groups than group by group (not pointwise?): '
                              'tabulating group by group', RuntimeWarning)
                self._batch_tab = False
                return [None] * len(triples)
        return tabs

    def load_group(self, X, XV, BX, domain, n_glob=None, nb_glob=None, into=None, shared_grid_t0=None, tab=None, verify=True,
                   hints=None, grids=None):
        """Prepare one group.  The user's callables (h, f, g, func_w, a, b) are evaluated on the device the given
        tensors live on and only their results are uploaded: pass the loader's host tensors to tabulate exactly like
        the reference's CPU path, or device tensors to tabulate on the GPU (float32 transcendental functions then
        differ from the host's in the last bit).  `into`: a Group of the same shapes to refill in place.
        `shared_grid_t0`: the caller built X, XV and BX from ONE time grid whose first time is this value (compact cube
        samples): the checks that would otherwise read the device tensors back (four host syncs) are skipped.
        `hints` (list domains, sampling.Comb_loader.pack): the same facts per group, read off the loader's HOST copies --
        `shared_times` (all paths of the group share one time column), `same_grid` (the boundary group sits on the
        interior group's grid).
        `n_glob`, `nb_glob` (several ranks): X, XV, BX are this rank's SHARE of a group of that many interior / boundary paths --
        the sub-steps of the group then run the exchange steps of dist.py; None: the whole group is here (one process, or a
        group every rank computes in full, dist.World.replicated).  A share may be empty ([0, L, d+1]: a group with fewer paths
        than ranks); `grids` = (time column of the WHOLE group's first interior path, of its first boundary path): the reference
        integrates every path of a group on its first path's grid (src/model.py:92: inputs[0, :, 0]), which an empty share does
        not hold and a later share holds a different one of where the paths of a group enter at their own times."""
        dev, d = self.dev, self.d
        sharded = n_glob is not None and self.world is not None
        X, XV = X.detach(), XV.detach()
        BX = BX.detach() if BX is not None else None
        N, L = X.shape[0], X.shape[1]
        Nb = BX.shape[0] if BX is not None else 0
        if XV.shape[1] != L or XV.shape[0] != N:
            raise XnwanError('u- and v-samples of a group must have the same shape')
        if (N == 0 or (BX is not None and Nb == 0)) and not (sharded and grids is not None):
            raise XnwanError("a group without interior or boundary paths only exists as a rank's share of a sharded group (n_glob, grids)")
        S = {}
        S['t'] = _d64(grids[0] if grids is not None else X[0, :, 0], dev).contiguous()
        S['xT'] = _d64(X[:, 0, 1:], dev).t().contiguous()
        S['xvT'] = _d64(XV[:, 0, 1:], dev).t().contiguous()
        # the test network is pointwise on XV: when the paths of a group do not share one time column (late-entry groups
        # of the hourglass: every path has its own entry time at l = 0) it runs in point mode on all L*N points
        S['tpp'] = S['tpp0'] = S['xvT_pts'] = None
        # (a share of a sharded group is compared with the WHOLE group's first path, whose grid G.t is: `grids`)
        col0 = XV[:1, :, 0] if grids is None else grids[0].to(XV.device).view(1, -1)
        if shared_grid_t0 is None and not (hints['shared_times'] if hints is not None else bool(torch.all(XV[:, :, 0] == col0))):
            S['tpp'] = _d64(XV[:, :, 0], dev).t().contiguous().reshape(-1)              # time-major: p = l*N + n
            S['tpp0'] = _d64(XV[:, 0, 0], dev).contiguous()
            S['xvT_pts'] = S['xvT'].unsqueeze(1).expand(d, L, N).reshape(d, L * N).contiguous()
        # start values and their x-gradient (the h -> y0 path of nabla_x u, src/model.py:95)
        if tab is not None:                   # (tabulate_sample: evaluated once for all groups of the sample)
            starts_T0 = bool(tab['starts_T0'])
            S['start'] = _d64(tab['start'], dev).reshape(-1).contiguous()
            S['ghT'] = _d64(tab['gh'], dev).t().contiguous()
            S['h'] = _d64(tab['h'], dev).reshape(-1).contiguous()
            S['f'] = _to_LN(tab['f'], dev)
            w, gw = tab['w'], tab['gw']
        elif N == 0:
            # an empty share: the user's callables are not asked about no points (max(), indexing ... of an empty tensor)
            starts_T0 = float(hints['t0'] if hints is not None else (shared_grid_t0 if shared_grid_t0 is not None else grids[0][0])) == self.setup['T0']
            z = lambda *shape: torch.zeros(*shape, dtype=F64, device=dev)  # noqa: E731
            S['start'], S['ghT'], S['h'], S['f'] = z(0), z(d, 0), z(0), z(L, 0)
            Lw = 1 if getattr(domain, 'time_independent', False) else L
            w, gw = z(0, Lw), z(0, Lw, d + 1)
        else:
            X0 = X[:, 0, :].clone().requires_grad_(True)
            t_first = grids[0][0] if grids is not None else X[0, 0, 0]
            starts_T0 = (float(t_first) if shared_grid_t0 is None else float(shared_grid_t0)) == self.setup['T0']
            s = self.funcs['h'](X0) if starts_T0 else self.funcs['g'](X0.unsqueeze(1)).reshape(-1)
            S['start'] = _d64(s.detach(), dev).reshape(-1).contiguous()
            if s.requires_grad:
                S['ghT'] = torch.autograd.grad(s.sum(), X0)[0][:, 1:].to(dev).to(F64).t().contiguous()
            else:
                S['ghT'] = torch.zeros(d, N, dtype=F64, device=dev)
            # (a group that starts at T0: h(X[:, 0, :]) is the start value itself)
            S['h'] = S['start'] if starts_T0 else self.funcs['h'](X[:, 0, :]).detach().to(dev).to(F64).reshape(-1).contiguous()
            S['f'] = _to_LN(self.funcs['f'](X), dev)
            # distance weight on the v-sample and its gradient (nabla phi = w nabla v + v nabla w, src/loss.py:51-63): in
            # closed form where the domain offers it (identical to autograd's, ties included), on the first time slice only
            # where w does not depend on time
            XVw = XV[:, :1] if getattr(domain, 'time_independent', False) else XV
            if hasattr(domain, 'func_w_grad'):
                w, gw = domain.func_w_grad(XVw)
            else:
                XVl = XVw.clone().requires_grad_(True)
                w = domain.func_w(XVl)
                gw = torch.autograd.grad(w.sum(), XVl)[0] if w.requires_grad else torch.zeros_like(XVl)
        if getattr(domain, 'time_independent', False):
            S['w'] = _d64(w[:, 0].detach(), dev).contiguous()
            S['wt'] = None
        else:
            S['w'] = _to_LN(w, dev)
            S['wt'] = _to_LN(gw[:, :, 0], dev)
        S['w0'] = _d64(w[:, 0].detach(), dev).contiguous()
        S['gwx0T'] = _d64(gw[:, 0, 1:], dev).t().contiguous()
        S['xbT'] = S['start_b'] = S['g'] = S['tb'] = None
        Lb, same_grid, b_T0 = 0, True, False
        if BX is not None:
            Lb = BX.shape[1]
            S['tb'] = _d64(grids[1] if grids is not None else BX[0, :, 0], dev).contiguous()
            same_grid = Lb == L and (shared_grid_t0 is not None or (hints['same_grid'] if hints is not None else bool(torch.equal(S['tb'], S['t']))))
            S['xbT'] = _d64(BX[:, 0, 1:], dev).t().contiguous()
            if tab is not None:
                b_T0 = bool(tab['b_T0'])
                S['start_b'] = _d64(tab['start_b'], dev).reshape(-1).contiguous()
                S['g'] = _to_LN(tab['g'], dev)
            elif Nb == 0:
                b_T0 = float(hints['tb0'] if hints is not None else (shared_grid_t0 if shared_grid_t0 is not None else grids[1][0])) == self.setup['T0']
                S['start_b'], S['g'] = torch.zeros(0, dtype=F64, device=dev), torch.zeros(Lb, 0, dtype=F64, device=dev)
            else:
                tb_first = grids[1][0] if grids is not None else BX[0, 0, 0]
                b_T0 = (float(tb_first) if shared_grid_t0 is None else float(shared_grid_t0)) == self.setup['T0']
                sb = self.funcs['h'](BX[:, 0, :]) if b_T0 else self.funcs['g'](BX[:, 0, :].unsqueeze(1)).reshape(-1)
                S['start_b'] = _d64(sb.detach(), dev).reshape(-1).contiguous()
                S['g'] = _to_LN(self.funcs['g'](BX), dev)
        st = self.structure
        S['X'] = X.to(dev) if st.c_kappa is None else None      # only read by a general reaction callable c(u, t, x)
        S['A0'] = S['B0'] = None
        amode = 0
        if not st.a_identity and N > 0:
            S['A0'], amode = self._tabulate_a(X[:, :1, :])     # [d,d,N] / [d,N] (diagonal) / [d,d] (constant) at time index 0
        if not st.b_zero and N > 0:
            S['B0'] = self._tabulate_b(X[:, :1, :])            # [d, N]
        if self.verify_structure and verify:
            ver = getattr(into, 'sample_version', 0) if into is not None else 0
            if ver % self.verify_every == 0 and N > 0:   # the first sample of a group and every verify_every-th refill after it
                self._check_structure(X, ver // self.verify_every)
        vol = float(domain.V())
        nglob = float(n_glob if n_glob is not None else N)
        nbglob = float(nb_glob if nb_glob is not None else max(Nb, 1))
        # single-slice groups at T0: the reference's [N,N] broadcasts in factorised form.  The sums over m of the pair
        # tables only involve the SAMPLE (h, f, g), so they are formed here, once per sample:
        #   mean_nm (u_n - h_m)^2 = mean_n (u_n - mean h)^2 + var h          (src/loss.py:79; :84 likewise with g)
        #   sum_mn f_m phi_n      = N sum_n mean(f) phi_n                    (src/loss.py:70)
        pair_i = self.pairwise_single_slice and L == 1 and starts_T0
        pair_b = self.pairwise_single_slice and (nb_glob if sharded else Nb) > 0 and Lb == 1 and b_T0
        S['href'] = None
        init_off = bdry_off = 0.0
        if pair_i and not st.b_zero:
            raise XnwanError('func_b != 0 on a single-slice group at T0: the reference sums that term with np.sum over a list of '
                             'tensors (src/loss.py:69), whose result on its [N,N] broadcast depends on the numpy version -- not '
                             'reproducible; set XW_ELEMENTWISE_SINGLE_SLICE=1 for the elementwise form')
        if pair_i or pair_b:
            zero = torch.zeros((), dtype=F64, device=dev)
            gsum, gsq = (S['g'].sum(), (S['g'] ** 2).sum()) if pair_b else (zero, zero)
            stats = torch.stack([S['h'].sum(), (S['h'] ** 2).sum(), S['f'].sum(), gsum, gsq]).contiguous()
            if sharded:
                self.world.all_reduce(stats)                 # the means are over ALL paths of the group, not this rank's share
            sh, shh, sf, sg, sgg = stats.tolist()
            if pair_i:
                S['href'] = torch.full((N,), sh / nglob, dtype=F64, device=dev)
                S['f'] = torch.full_like(S['f'], sf / nglob)
                init_off = shh / nglob - (sh / nglob) ** 2
            if pair_b:
                S['g'] = torch.full_like(S['g'], sg / nbglob)
                bdry_off = sgg / nbglob - (sg / nbglob) ** 2
        # `sharded`: the sums of this group are partial sums (exchange steps in its sub-steps); `has_bdry`: the GROUP has boundary
        # paths, whether or not this rank holds any (which parameters Adam skips must not depend on the rank, _gen_back)
        pair_state = dict(pair_i=pair_i, pair_b=pair_b, init_off=init_off, bdry_off=bdry_off, s3_scale=nglob if pair_i else 1.0,
                          sharded=sharded, has_bdry=BX is not None and (nb_glob if sharded else Nb) > 0)
        if into is not None:
            G = into
            same = (G.N, G.L, G.Nb, G.Lb, G.same_grid, G.Vol, G.Nglob, G.Nbglob, G.pair_i, G.pair_b, G.amode, G.sharded) == (
                N, L, Nb, Lb, same_grid, vol, nglob, nbglob, pair_i, pair_b, amode, sharded) and all(
                (getattr(G, k) is None) == (S[k] is None) and (S[k] is None or getattr(G, k).shape == S[k].shape)
                for k in Group.SAMPLE_FIELDS)
            if same:
                for k in Group.SAMPLE_FIELDS:
                    if S[k] is not None:
                        getattr(G, k).copy_(S[k])
                G.domain = domain
                G.__dict__.update(pair_state)
                G.sample_version += 1
                return G
        G = Group()
        for k in Group.SAMPLE_FIELDS:
            setattr(G, k, S[k])
        G.domain, G.N, G.L, G.Nb, G.Vol, G.Nglob, G.Nbglob = domain, N, L, Nb, vol, nglob, nbglob
        G.Lb, G.same_grid, G.amode = Lb, same_grid, amode
        G.__dict__.update(pair_state)
        # work buffers
        # XW_POISON=1 (debugging): work buffers start as NaN instead of whatever the allocator hands out, so that a kernel
        # reading a slot nobody wrote shows up as NaN in the results instead of as a stale, plausible number
        self._work_buffers(G, [])
        G.graphs = {}
        # (a group of a list domain changes shape with every sample and is built anew each time: the count of samples it has seen
        #  is carried over, or the periodic structure guard above -- two read-backs -- would run on EVERY sample)
        G.sample_version = into.sample_version + 1 if into is not None else 0
        return G

    def load_groups_packed(self, shards, hints, domain, cache, big):
        """load_group for ALL groups of a list-domain sample at once (time-varying balls: 11-20 groups): every sample field of
        every group is a strided view of the uploaded sample or of a table tabulate_sample has just filled for the whole
        sample, so the fields are regions of each group's one allocation (Group._arena) and ONE gather launch
        (kernels.gather_fields, a ~350-row table uploaded once) fills them -- instead of ~45 tensor operations per group.
        Same values bit for bit (copies), same Group attributes.  Returns None when the sample is not of that kind (anything
        but float64 device tensors, general a / b / c, no batched tabulation): the caller goes group by group.
        `shards`: (X, XV, BX, n_glob, nb_glob, grids) per group as in load_group -- with several ranks this rank's shares (possibly
        empty) of the groups that are sharded, the whole of those that are replicated."""
        tc = getattr(self, '_tab_cat', None)
        st = self.structure
        if (tc is None or not (st.a_identity and st.b_zero and st.c_kappa is not None)
                or getattr(domain, 'time_independent', False) or any(h is None for h in hints)):
            return None
        srcs = [tc[k] for k in ('f', 'g', 'w', 'gw', 'start', 'gh', 'h', 'start_b')]
        if any(t.dtype != F64 or not t.is_cuda for t in srcs) or any(t.dtype != F64 or not t.is_cuda or t_[0] is not t_[1]
                                                                       for t_ in shards for t in t_[:3]):
            return None
        d, dev, T0 = self.d, self.dev, self.setup['T0']
        f_cat, g_cat, w_cat, gw_cat, start, gh, hv, sb = srcs
        if f_cat.stride(0) != 1 or g_cat.stride(0) != 1 or w_cat.stride(0) != 1:
            return None
        g0, g1 = gw_cat.stride()
        h0, h1 = gh.stride()
        rows, groups, total = [], [], 0
        oN = oNL = oNb = oNbL = 0
        vol = float(domain.V())
        for k, ((X, XV, BX, n_glob, nb_glob, grids), hn) in enumerate(zip(shards, hints)):
            N, L, Nb, Lb = X.shape[0], X.shape[1], BX.shape[0], BX.shape[1]
            sharded = n_glob is not None and self.world is not None
            shared = bool(hn['shared_times'])
            same_grid = Lb == L and bool(hn['same_grid'])
            starts_T0, b_T0 = bool(tc['at0'][k]), bool(tc['bat0'][k])
            G = Group()
            G.domain, G.N, G.L, G.Nb, G.Vol = domain, N, L, Nb, vol
            G.Nglob, G.Nbglob = float(n_glob if n_glob is not None else N), float(nb_glob if nb_glob is not None else max(Nb, 1))
            G.Lb, G.same_grid, G.amode = Lb, same_grid, 0
            fields = [('t', (L,)), ('xT', (d, N))]
            if not shared:
                fields += [('tpp', (L * N,)), ('tpp0', (N,)), ('xvT_pts', (d, L * N))]
            fields += [('start', (N,)), ('ghT', (d, N)), ('h', (N,)), ('f', (L, N)), ('w', (L, N)), ('wt', (L, N)), ('w0', (N,)), ('gwx0T', (d, N)),
                       ('tb', (Lb,)), ('xbT', (d, Nb)), ('start_b', (Nb,)), ('g', (Lb, Nb))]
            self._work_buffers(G, fields)
            G._lazy['xvT'] = G._lazy['xT']           # (the v sample of a list domain is the u sample)
            if shared:
                G.tpp = G.tpp0 = G.xvT_pts = None
            G.href = G.X = G.A0 = G.B0 = None
            base = G._arena.data_ptr()
            x0, x1, x2 = X.stride()
            b0, b1, b2 = BX.stride()
            xp, bp = X.data_ptr(), BX.data_ptr()
            # (the grids are those of the WHOLE group's first paths: load_group)
            t_src = (xp, x1) if grids is None else (grids[0].data_ptr(), grids[0].stride(0))
            tb_src = (bp, b1) if grids is None else (grids[1].data_ptr(), grids[1].stride(0))
            src = {'t': (t_src[0], 1, 1, L, 0, 0, t_src[1]), 'xT': (xp + 8 * x2, 1, d, N, 0, x2, x0),
                   'tpp': (xp, 1, L, N, 0, x1, x0), 'tpp0': (xp, 1, 1, N, 0, 0, x0), 'xvT_pts': (xp + 8 * x2, d, L, N, x2, 0, x0),
                   'start': (start.data_ptr() + 8 * oN * start.stride(0), 1, 1, N, 0, 0, start.stride(0)),
                   'ghT': (gh.data_ptr() + 8 * oN * h0, 1, d, N, 0, h1, h0),
                   'h': (hv.data_ptr() + 8 * oN * hv.stride(0), 1, 1, N, 0, 0, hv.stride(0)),
                   'f': (f_cat.data_ptr() + 8 * oNL, 1, L, N, 0, 1, L),
                   'w': (w_cat.data_ptr() + 8 * oNL, 1, L, N, 0, 1, L),
                   'wt': (gw_cat.data_ptr() + 8 * oNL * g0, 1, L, N, 0, g0, L * g0),
                   'w0': (w_cat.data_ptr() + 8 * oNL, 1, 1, N, 0, 0, L),
                   'gwx0T': (gw_cat.data_ptr() + 8 * (oNL * g0 + g1), 1, d, N, 0, g1, L * g0),
                   'tb': (tb_src[0], 1, 1, Lb, 0, 0, tb_src[1]), 'xbT': (bp + 8 * b2, 1, d, Nb, 0, b2, b0),
                   'start_b': (sb.data_ptr() + 8 * oNb * sb.stride(0), 1, 1, Nb, 0, 0, sb.stride(0)),
                   'g': (g_cat.data_ptr() + 8 * oNbL, 1, Lb, Nb, 0, 1, Lb)}
            for name, shape in fields:
                a, n0_, n1_, n2_, s0_, s1_, s2_ = src[name]
                if n0_ * n1_ * n2_ == 0:        # (an empty share: nothing to gather)
                    continue
                rows.append((a, base + 8 * G._lazy[name][0], n0_, n1_, n2_, s0_, s1_, s2_, total))
                total += n0_ * n1_ * n2_
            has_bdry = (nb_glob if sharded else Nb) > 0
            pair_i = self.pairwise_single_slice and L == 1 and starts_T0
            pair_b = self.pairwise_single_slice and has_bdry and Lb == 1 and b_T0
            G.__dict__.update(dict(pair_i=pair_i, pair_b=pair_b, init_off=0.0, bdry_off=0.0, s3_scale=G.Nglob if pair_i else 1.0,
                                   sharded=sharded, has_bdry=has_bdry))
            old = cache[k] if k < len(cache) else None
            G.graphs = {}
            ver = old.sample_version if old is not None else 0
            G.sample_version = ver + 1 if old is not None else 0
            if self.verify_structure and k == big and ver % self.verify_every == 0:      # (as load_group: the first sample and every 16th)
                self._check_structure(X, ver // self.verify_every)
            groups.append(G)
            oN, oNL, oNb, oNbL = oN + N, oNL + N * L, oNb + Nb, oNbL + Nb * Lb
        if len(rows) > KN.GATHER_ROWS:
            return None
        table = torch.zeros(KN.GATHER_ROWS, 9, dtype=torch.int64)
        table[:len(rows)] = torch.tensor(rows, dtype=torch.int64)
        from .sampling import _PIN_POOL
        pinned = _PIN_POOL.stage(table)
        tdev = pinned.to(dev, non_blocking=True)
        _PIN_POOL.uploaded(pinned, torch.cuda.current_stream(dev))
        KN.gather_fields(tdev, len(rows), total)
        # single-slice groups at T0: the reference's [N, N] broadcasts in factorised form (load_group); these read five sums back
        for G in groups:
            if G.pair_i and not st.b_zero:
                raise XnwanError('func_b != 0 on a single-slice group at T0 (see load_group)')
            if G.pair_i or G.pair_b:
                zero = torch.zeros((), dtype=F64, device=dev)
                gsum, gsq = (G.g.sum(), (G.g ** 2).sum()) if G.pair_b else (zero, zero)
                stats = torch.stack([G.h.sum(), (G.h ** 2).sum(), G.f.sum(), gsum, gsq]).contiguous()
                if G.sharded:
                    self.world.all_reduce(stats)             # (the means are over ALL paths of the group: load_group)
                sh, shh, sf, sg, sgg = stats.tolist()
                if G.pair_i:
                    G.href = torch.full((G.N,), sh / G.Nglob, dtype=F64, device=dev)
                    G.f.fill_(sf / G.Nglob)
                    G.init_off = shh / G.Nglob - (sh / G.Nglob) ** 2
                if G.pair_b:
                    G.g.fill_(sg / G.Nbglob)
                    G.bdry_off = sgg / G.Nbglob - (sg / G.Nbglob) ** 2
        return groups

    def _work_buffers(self, G, fields):
        """every buffer the sub-steps of G need, as regions of ONE allocation (Group._arena / _lazy), behind the regions `fields`
        = [(name, shape)] the caller fills itself (load_groups_packed: the sample fields)"""
        dev, d, N, L, Nb, Lb = self.dev, self.d, G.N, G.L, G.Nb, G.Lb
        poison = self.options.poison
        H = self.H
        # stage activations of every step, written by the forward, read back by the sweeps (183 MB at N = 4096, L = 32)
        ar = KN.ode_act_rows(self.method, H, self.K, self.m) if self.keep_activations and not self.adjoint else 0
        # layer inputs of the test network at every point, stored by its forward in the discriminator sub-step and read
        # back by its backward (524 MB at 131072 points)
        # (only the reference's width and depth have a recomputing reverse kernel: everything else always runs from the record)
        keep_v = self.keep_activations or not KN.disc_recompute(self.W, self.q)
        G.ns_u = KN.ode_bwd_slabs(N)
        G.ns_b = KN.ode_bwd_slabs(Nb) if Nb else 0
        nw = KN.reduce_work_size()
        plan = list(fields) + [('u', (L, N)), ('Y', (L, H, N)), ('v', (L, N)), ('vt', (L, N)), ('gxv', (d, N)), ('gtv', (N,)), ('gx', (d, N)), ('gs', (N,)),
                ('vbar', (L, N)), ('s3x', (N,)),
                ('slabA', (G.ns_u + G.ns_b, self.Pu)),         # sweep with cotangent A (interior) + the boundary sweep
                ('slabB', (G.ns_u, self.Pu)),                  # sweep with cotangent B = dI/du
                ('slab_v', (KN.disc_bwd_slabs(N, L), self.Pv)),
                ('work_i', (nw,)), ('work_b', (nw,))]          # scratch of the deterministic grid sums (interior / boundary run concurrently): zeroed below
        if ar:
            plan.append(('act', (max(L - 1, 1), ar, KN.ode_act_cols(N))))
            if Nb:
                plan.append(('act_b', (max(Lb - 1, 1), ar, KN.ode_act_cols(Nb))))
        if keep_v:
            plan.append(('vact', (KN.disc_act_rows(self.W, self.q), KN.disc_act_cols(L * N))))
        if N and d >= self.xproj_min_d:
            plan.append(('xproj', (KN.disc_xproj_rows(self.W), N)))   # Vin[:, 1..d] x + Vin.b per path (_launch_test_net_here)
        if Nb:
            plan += [('ub', (Lb, Nb)), ('Yb', (Lb, H, Nb))]
        lazy, off = {}, 0
        for name, shape in plan:
            lazy[name] = (off, shape)
            off += -(-math.prod(shape) // 64) * 64           # (regions start on 512-byte boundaries, like allocations of their own)
        G._arena = torch.full((off,), float('nan'), dtype=F64, device=dev) if poison else torch.empty(off, dtype=F64, device=dev)
        G._lazy = lazy
        w0 = lazy['work_i'][0]
        G._arena[w0:lazy['work_b'][0] + nw].zero_()
        G.c = G.cp = None
        if not ar:
            G.act = G.act_b = None
        elif not Nb:
            G.act_b = None
        if not keep_v:
            G.vact = None

    def refill_compact(self, G, comp, domain, n_glob=None, nb_glob=None):
        """load_group(..., into=G) for a COMPACT sample of a domain whose paths are vertical lines over one shared grid
        (times [L], x_u [N, d], x_v [N, d], x_b [N_b, d]; host tensors, page-locked ones upload asynchronously): the
        sample goes into four static device buffers and everything load_group does with it on the device -- the path
        tensors, the callables h (with its gradient), f, g, the domain's weight, the transposes and conversions into the
        group's sample fields, ~110 small kernels -- is captured ONCE per shape into a HIP graph and replayed: an outer
        iteration of train() at the headline size spent 1.0 of its 2.4 ms of host time issuing them one by one.  Same
        rule as the sub-step graphs (_run): a callable that cannot be captured (it syncs with the host) makes this
        segment eager, with a warning.  The guard on the probed coefficient structure stays outside the graph."""
        times = comp[0]
        st = G.__dict__.get('_refill_in')
        if st is None or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(st, comp)):
            st = G._refill_in = [torch.empty(tuple(c.shape), dtype=c.dtype, device=self.dev) for c in comp]
            G.graphs = {k: v for k, v in G.graphs.items() if not k.startswith('refill')}
        for dst, src in zip(st, comp):
            dst.copy_(src, non_blocking=True)
        _PIN_POOL.uploaded_all(comp, torch.cuda.current_stream(self.dev))
        t0 = float(times[0])                                   # (host tensor: no device read-back)
        td, du, dv, db = st
        ver = G.sample_version
        if self.verify_structure and ver % self.verify_every == 0:
            self._check_structure(_paths(td, du), ver // self.verify_every)

        lean = (self.structure.a_identity and self.structure.b_zero and self.structure.c_kappa is not None
                and getattr(domain, 'time_independent', False) and hasattr(domain, 'func_w_grad') and G.Nb > 0 and G.same_grid
                and not (G.pair_i or G.pair_b) and G.tpp is None
                and (G.N, G.L, G.Nb, G.Lb) == (du.shape[0], td.shape[0], db.shape[0], td.shape[0])
                and float(domain.V()) == G.Vol and n_glob in (None, G.Nglob) and nb_glob in (None, G.Nbglob))

        def body(_G):
            if lean:
                self._refill_fields(G, td, du, dv, db, domain, t0)
                return
            out = self.load_group(_paths(td, du), _paths(td, dv), _paths(td, db), domain, n_glob, nb_glob, into=G, shared_grid_t0=t0,
                                  verify=False)
            if out is not G:
                raise XnwanError('refill_compact: the sample does not have the shapes of the group it refills')
        # (what the captured body bakes in besides the buffers: the grid's first time, the domain's class and extent, which of
        #  the two bodies it is and the global path counts)
        key = 'refill_%r_%s_%r_%r_%r_%d_%r_%r' % (t0, type(domain).__name__, float(domain.V()), getattr(domain, 'top', None),
                                                  getattr(domain, 'bot', None), int(lean), n_glob, nb_glob)
        if key not in G.graphs and sum(1 for k in G.graphs if k.startswith('refill')) >= self.refill_variants:
            # a domain whose first time or extent changes with every sample would capture a graph per outer iteration (graphs are
            # never destroyed, _KEPT_GRAPHS): beyond a few variants per group the refill runs eagerly
            if not G.__dict__.get('_refill_cap_warned'):
                import warnings
                G._refill_cap_warned = True
                warnings.warn('more than %d variants of the refill graph for one group (first time / extent / path counts change from '
                              'sample to sample): further variants run as eager launches' % self.refill_variants, RuntimeWarning, stacklevel=2)
            G.graphs[key] = False
        self._run(G, key, body, scratch=True)
        # (host-side bookkeeping of load_group: done at capture time only, so it is set here on every path)
        G.domain, G.sample_version = domain, ver + 1
        return G

    def _refill_fields(self, G, td, du, dv, db, domain, t0):
        """What load_group(paths(td, du), paths(td, dv), paths(td, db), into=G, shared_grid_t0=t0) writes into the sample
        fields of a group of vertical paths over ONE grid, on a time-independent domain with the fused coefficient structure
        (a = identity, b = 0, c = kappa u) -- the same values bit for bit (the same callables on the same tensors; every
        conversion is an exact float32 -> float64 widening or a transpose), written straight into the fields: a widening, a
        transpose and the copy into the field are ONE strided copy here and three kernels there (~110 -> ~85 per sample).
        The four independent chains -- f on the interior paths | the start values h with their gradient | the domain's
        weight | the boundary paths -- run on the engine's side streams: captured (refill_compact), that makes them
        parallel branches of the graph, and the GPU has nothing else to do between two outer iterations."""
        T0, fn = self.setup['T0'], self.funcs
        at0 = float(t0) == T0
        e0 = self._mark()
        with self._side(1, e0):                       # start values and their x-gradient (the h -> y0 path of nabla_x u)
            X0 = _paths(td[:1], du)[:, 0, :].requires_grad_(True)
            s_ = fn['h'](X0) if at0 else fn['g'](X0.unsqueeze(1)).reshape(-1)
            G.start.copy_(s_.detach().reshape(-1))
            if s_.requires_grad:
                G.ghT.copy_(torch.autograd.grad(s_.sum(), X0)[0][:, 1:].t())
            else:
                G.ghT.zero_()
            # (a group that starts at T0: h on its first points IS the start value)
            G.h.copy_(G.start if at0 else fn['h'](X0.detach()).reshape(-1))
            e1 = self._mark()
        with self._side(2, e0):                       # distance weight on the v-sample, first time slice (time-independent)
            stock = type(domain).func_w_grad is Hypercube.func_w_grad and type(domain).func_w is Hypercube.func_w
            if stock and dv.dtype == F32 and dv.is_contiguous():
                # the stock hypercube: its weight, the gradient and the transposed points in ONE launch (xw_cube_weight: same
                # float32 arithmetic, same tie rules as the tensor formulation below, which takes 29)
                KN.cube_weight(dv, domain.top, domain.bot, G.w, G.gwx0T, w0=G.w0, xT=G.xvT)
            else:
                w, gw = domain.func_w_grad(_paths(td[:1], dv))
                G.w.copy_(w[:, 0].detach())
                G.w0.copy_(w[:, 0].detach())
                G.gwx0T.copy_(gw[:, 0, 1:].t())
                G.xvT.copy_(dv.t())
            e2 = self._mark()
        with self._side(3, e0):                       # boundary paths
            BX = _paths(td, db)
            sb = fn['h'](BX[:, 0, :]) if at0 else fn['g'](BX[:, 0, :].unsqueeze(1)).reshape(-1)
            G.start_b.copy_(sb.detach().reshape(-1))
            G.g.copy_(fn['g'](BX).detach().t())
            G.xbT.copy_(db.t())
            G.tb.copy_(td)
            e3 = self._mark()
        G.t.copy_(td)
        G.xT.copy_(du.t())
        G.f.copy_(fn['f'](_paths(td, du)).detach().t())
        self._join(e1, e2, e3)

    # ------------------------------------------------------------------------------------------------------------
    # building blocks
    # ------------------------------------------------------------------------------------------------------------
    def _side(self, i, *events):
        """context: run on side stream i after `events` (falls back to the current stream when streams are off)"""
        if not self.use_streams:
            return _NOSTREAM                      # one stream: program order is the dependency (no contexts, no events)
        st = self.streams[i]
        for ev in events:
            if ev is not None:
                st.wait_event(ev)
        return torch.cuda.stream(st)

    def _mark(self):
        return torch.cuda.current_stream().record_event() if self.use_streams else None

    def _join(self, *events):
        if self.use_streams:
            cur = torch.cuda.current_stream()
            for ev in events:
                if ev is not None:
                    cur.wait_event(ev)

    def _launch_test_net_here(self, G, blocks=None):
        """test network on the CURRENT stream (the sub-step's critical chain): v, dv/dt at all points; nabla_x v at the first
        time index rides along in the same launch (fused reverse chain); optionally the record of layer inputs for the
        backward (decided in _v_fresh, outside the captured code)"""
        ph, blocks = self.phi.data, blocks or self.v_blocks
        act = G.vact if getattr(G, 'vact_valid', False) else None
        if G.tpp is not None:
            KN.disc_fwd(G.xvT_pts, None, ph, self.W, self.q, tpp=G.tpp, v=G.v.view(1, -1), vt=G.vt.view(1, -1),
                        gxv=G.gxv, gtv=G.gtv, ngrad=G.N, max_blocks=blocks, act=act)
        else:
            # (the input layer's spatial columns do not move along a vertical path: applied once per path by a small launch in
            #  front, csrc/xw_disc.hip k_disc_xproj -- the main launch then loads its row of that table instead of x)
            xp = KN.disc_xproj(G.xvT, ph, self.W, out=G.xproj) if 'xproj' in G._lazy else None
            KN.disc_fwd(G.xvT, G.t, ph, self.W, self.q, v=G.v, vt=G.vt, gxv=G.gxv, gtv=G.gtv, ngrad=G.N,
                        max_blocks=blocks, act=act, xproj=xp)

    def _reaction(self, G):
        """c(u, t, x): linear fast path, or the user's callable differentiated by autograd (not graph-capturable)"""
        ck = self.structure.c_kappa
        if ck is None and G.pair_i:
            # single-slice group at T0: the reference hands its callable u as [N, 1] there (src/model.py:89-91), not [N, 1, 1], and takes
            # `c.squeeze() * u.squeeze() * phi.squeeze()` summed over ALL its entries (src/loss.py:70).  A callable that mixes u with a
            # slice of X that kept its last axis ([N, 1, 1]) thereby returns the TABLE c[m, n] = c(u_n, x_m): per path n the term is
            # sum_m c[m, n] u_n phi_n = N mean_m c[m, n] u_n phi_n -- the N is the one every term of such a group carries (s3_scale)
            N = G.N
            ul = G.u.t().detach().requires_grad_(True)                           # [N, 1]
            with torch.enable_grad():
                c = self.funcs['c'](G.X, ul)
                c = c.reshape(N, N) if c.numel() == N * N and N > 1 else c.reshape(-1)
                if c.dim() == 2:
                    if getattr(G, 'sharded', False):
                        raise XnwanError('func_c returns a table over all pairs of paths on a single-slice group (it mixes u [N, 1] with a slice of X that '
                                         'kept its last axis): that needs every path of the group on one rank -- run such groups replicated '
                                         '(XW_REPLICATE_BELOW >= their size) or on one GPU')
                    c = c.mean(0)
                elif c.numel() != N:
                    raise XnwanError('func_c on a single-slice group returned %d values for %d paths' % (c.numel(), N))
                cp = torch.autograd.grad(c.sum(), ul)[0] if c.requires_grad else torch.zeros_like(ul)
            if G.c is None:
                G.c, G.cp = torch.empty_like(G.u), torch.empty_like(G.u)
            G.c.copy_(c.detach().reshape(1, N))
            G.cp.copy_(cp.detach().reshape(1, N))
            ck = 0.0
        elif ck is None:
            ul = G.u.t().unsqueeze(2).detach().requires_grad_(True)
            with torch.enable_grad():
                c = self.funcs['c'](G.X, ul)
                cp = torch.autograd.grad(c.sum(), ul)[0] if c.requires_grad else torch.zeros_like(ul)
            if G.c is None:
                G.c, G.cp = torch.empty_like(G.u), torch.empty_like(G.u)     # (persistent: captured graphs keep reading them)
            G.c.copy_(c.detach().squeeze(2).t())
            G.cp.copy_(cp.detach().squeeze(2).t())
            ck = 0.0
        G.ck = ck

    # ------------------------------------------------------------------------------------------------------------
    # one C-ABI call per group sub-step (xw_substep_gen / xw_substep_disc, csrc/xw_substep.hip) for the groups that run
    # eagerly -- the 11-20 groups per sample of the list domains, new shapes every sample: ~15 launches per group and
    # sub-step issued from Python one by one cost more host time than the GPU needs to run them
    # ------------------------------------------------------------------------------------------------------------
    def _runner_state(self):
        """XwSolverState of this engine (pointers to buffers that live as long as the engine)"""
        from ._lib import XwSolverState
        st = getattr(self, '_xw_state', None)
        if st is None:
            st = self._xw_state = XwSolverState()
            st.method, st.H, st.K, st.m, st.W, st.q, st.Pu, st.Pv = self.method, self.H, self.K, self.m, self.W, self.q, self.Pu, self.Pv
            st.adjoint = 1 if self.adjoint else 0
            st.lag_lo, st.lag_hi = self.field_range
            st.beta1, st.beta2, st.eps = 0.9, 0.999, 1e-8
            st.theta, st.phi, st.scal = self.theta.data.data_ptr(), self.phi.data.data_ptr(), self.scal.data_ptr()
            st.grad_u, st.grad_v = self.grad_u.data_ptr(), self.grad_v.data_ptr()
            st.m_u, st.v_u, st.m_v, st.v_v = (self.adam_u['m'].data_ptr(), self.adam_u['v'].data_ptr(), self.adam_v['m'].data_ptr(),
                                              self.adam_v['v'].data_ptr())
            st.step_u, st.step_v, st.lag_u = self.adam_u['step'].data_ptr(), self.adam_v['step'].data_ptr(), self.adam_u['lag'].data_ptr()
            st.exchange = st.exchange_ctx = st.pack_u = None
            if self.world is not None:
                st.pack_u = self.pack_u.data_ptr()
                if self.world.comm is not None:            # RCCL: the library's own entry point, enqueued on the stream like the kernels
                    from ._lib import lib
                    import ctypes
                    st.exchange = ctypes.cast(lib.xw_allreduce, ctypes.c_void_p).value
                    st.exchange_ctx = self.world.comm.value if hasattr(self.world.comm, 'value') else self.world.comm
                else:                                      # rehearsal (gloo): a host-side stand-in with the same contract
                    self._exchange_cb = self._host_exchange()
                    st.exchange = ctypes_addr(self._exchange_cb)
        st.v_blocks, st.v_blocks_disc = self.v_blocks, self.v_blocks_disc
        st.alpha, st.pollution = float(self.alpha), float(self.pollution)
        st.lr_u, st.lr_v = float(self.config['u_rate']), float(self.config['v_rate'])
        return st

    def _host_exchange(self):
        """XwSolverState.exchange without RCCL (the `gloo` rehearsal: several ranks on one GPU, tests): the same contract as
        xw_allreduce -- in-place float64 sum of buf[count] over the ranks, in stream order -- through torch.distributed with the
        buffer staged on the host.  The runner only ever passes the engine's own exchange buffers."""
        from ._lib import EXCHANGE_FN
        bufs = {t.data_ptr(): t for t in (self.pack_u, self.scal, self.grad_v)}

        def exchange(buf, count, _ctx, _stream):
            try:
                self.world.all_reduce(bufs[buf][:count])
                return 0
            except BaseException as e:          # (an exception cannot cross the C frames: kept for the caller, reported as XW_E_COMM)
                self._exchange_error = e
                return -4
        return EXCHANGE_FN(exchange)

    def _runner_group(self, G):
        """XwGroup of a loaded group: pointers once per allocation, the per-sample scalars every time"""
        from ._lib import XwGroup
        xg = getattr(G, '_xw_group', None)
        if xg is None:
            xg = G._xw_group = XwGroup()
            p = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
            xg.N, xg.Nb, xg.L, xg.Lb, xg.d = G.N, G.Nb, G.L, G.Lb, self.d
            xg.same_grid, xg.w_per_point, xg.amode = int(bool(G.same_grid)), int(G.w.dim() == 2), int(G.amode)
            xg.ns_u, xg.ns_b = G.ns_u, G.ns_b
            for k in ('xT', 'xvT', 'xbT', 't', 'tb', 'tpp', 'xvT_pts', 'start', 'start_b', 'h', 'href', 'f', 'g', 'w', 'wt', 'w0', 'ghT',
                      'gwx0T', 'A0', 'B0', 'u', 'Y', 'act', 'act_b', 'v', 'vt', 'gxv', 'gtv', 'gx', 'gs', 'vbar', 's3x', 'vact',
                      'xproj', 'slabA', 'slabB', 'slab_v', 'work_i', 'work_b', 'ub', 'Yb'):
                setattr(xg, k, G.ptr(k))         # (addresses only: the work buffers' tensor views are made when Python reads them)
            nar = lambda jobs, **kw: self._narrow_ok(jobs, **kw)  # noqa: E731
            ji, jb = self._job(G, 'i'), (self._job(G, 'b') if G.Nb else None)
            joint = G.Nb and G.same_grid
            # (bit 2: the runner launches sweep A, the boundary sweep on the same grid and sweep B as ONE launch -- the tile count
            #  that decides is that launch's, as in _gen_front_compact; bit 4 is unused)
            bits = [nar([ji] + ([jb] if joint else []), alone=False, forward=True),
                    bool(jb) and nar([jb], alone=False, forward=True),
                    nar([ji] + ([jb] if joint else []) + [ji], alone=False),
                    bool(jb) and nar([jb], alone=False),
                    False,
                    nar([ji], alone=False, params=False),
                    nar([ji], alone=False, forward=True),
                    nar([ji], alone=False, params=False)]
            xg.narrow = sum(1 << i for i, b in enumerate(bits) if b)
        xg.sharded = int(bool(getattr(G, 'sharded', False)))
        xg.pair_i, xg.pair_b = int(bool(G.pair_i)), int(bool(G.pair_b))
        xg.Vol, xg.Nglob, xg.Nbglob, xg.s3_scale = float(G.Vol), float(G.Nglob), float(G.Nbglob), float(G.s3_scale)
        xg.init_off, xg.bdry_off, xg.ckappa = float(G.init_off), float(G.bdry_off), float(self.structure.c_kappa)
        xg.href = 0 if G.href is None else G.href.data_ptr()
        xg.c = xg.cp = 0
        return xg

    def _runner_ok(self, G):
        return (self.use_runner and self.structure.c_kappa is not None and G.c is None and not getattr(G, 'persistent', True))

    def _world_of(self, G):
        """the ranks that hold shares of G: None when the whole group is here (one process, or a group every rank computes in
        full -- dist.World.replicated -- with the single-process arithmetic and no exchange)"""
        return self.world if getattr(G, 'sharded', self.world is not None) else None

    def _field_seen(self, G):
        """has a group of this sub-iteration integrated the ODE yet?  (no: the field's parameters have no gradient, Adam skips
        them -- Engine.__init__.)  From facts every rank shares: `has_bdry` is the GROUP's, not this rank's possibly empty share"""
        touched = G.L > 1 or (getattr(G, 'has_bdry', G.Nb > 0) and G.Lb > 1)
        self._field_touched = touched or (self.accum_u is not None and self._field_touched)
        return self._field_touched

    def _narrow_ok(self, jobs, alone, forward=False, params=True):
        """narrow tiles (csrc/xw_ode_n4.h) for this stepper launch?  XW_NARROW: 0 never, 1 by size (default), 2 wherever the
        kernels exist; XW_NARROW_SET: which launches may (f forward, x sweeps without weight gradients, p sweeps with them)"""
        kind = 'f' if forward else ('p' if params else 'x')
        if self.narrow == '0' or kind not in self.narrow_set:
            return False
        if not forward and (self.adjoint or self.method > 1 or any(j.get('act') is None for j in jobs)):
            return False
        if self.narrow == '2':
            return True
        tiles = sum((j['xT'].shape[1] + 15) // 16 for j in jobs)
        # `alone`: nothing else runs beside this launch (the test network is reused / this is a lone evaluation) -- the narrow
        # forms then gain up to larger launches (narrow_tiles_alone), they need SIMDs to spare, not a short chain only
        return tiles <= (self.narrow_tiles_alone if alone else self.narrow_tiles)[kind]

    def _job(self, G, which, ubar=None, gslab=None, want_x=False):
        if which == 'i':
            j = dict(xT=G.xT, start=G.start, u=G.u, Y=G.Y, act=G.act, ubar=ubar, gslab=gslab)
            if want_x:
                j.update(gx=G.gx, gs=G.gs)
        else:
            j = dict(xT=G.xbT, start=G.start_b, u=G.ub, Y=G.Yb, act=G.act_b, ubar=ubar, gslab=gslab)
        return j

    def _contract(self, G, adam_state=None, with_bdry=False):
        """I, sum v^2, SSE_init from u, v, dv/dt and the two helper-backward gradients (src/loss.py:46-76).
        Single GPU: the sums are global, so the same launch also forms the loss values and advances the optimiser's
        counter (`adam_state`); with several GPUs that is done by KN.losses after the all-reduce."""
        fin = None
        if self._world_of(G) is None and adam_state is not None:
            fin = dict(Lb=G.Lb, Nbglob=G.Nbglob, alpha=self.alpha, step=adam_state['step'], init_off=G.init_off, bdry_off=G.bdry_off)
        pair = dict(href=G.href, s3_scale=G.s3_scale) if G.pair_i else None
        # generator sub-step: the boundary penalty's sum of squares (a loss value; the sweeps form its cotangent themselves) rides along
        bdry = dict(ub=G.ub, g=G.g) if (with_bdry and G.Nb) else None
        if G.A0 is None and G.B0 is None:
            KN.weak_partials(G.u, G.v, G.vt, G.w, G.f, G.h, G.Vol, G.Nglob, self.scal, G.work_i, c=G.c, ckappa=G.ck, wt=G.wt,
                             contract=dict(gx=G.gx, gs=G.gs, ghT=G.ghT, gxv=G.gxv, w0=G.w0, gwx0T=G.gwx0T), finalize=fin, pair=pair,
                             bdry=bdry)
            return
        # general a_ij / b_i: the l = 0 contraction as one streaming kernel over the tabulated slice (graph-capturable)
        KN.weak_contract_general(G.A0, G.amode, G.B0, G.gx, G.gs, G.ghT, G.gxv, G.w0, G.gwx0T, G.v[0], G.s3x)
        KN.weak_partials(G.u, G.v, G.vt, G.w, G.f, G.h, G.Vol, G.Nglob, self.scal, G.work_i, s3x=G.s3x, c=G.c, ckappa=G.ck,
                         wt=G.wt, finalize=fin, pair=pair, bdry=bdry)

    # ------------------------------------------------------------------------------------------------------------
    # generator sub-step (src/training.py:127-138)
    # ------------------------------------------------------------------------------------------------------------
    def _gen_all(self, G):
        """single GPU: the whole generator sub-step is one captured graph"""
        self._gen_front(G)
        self._gen_back(G)

    def _gen_front(self, G):
        """everything up to (not including) the exchange: leaves slabA, slabB and scal[0..3] complete.
        Kernel chains:  main   test network v, dv/dt and (fused) nabla_x v(t_0) -> cotangent B = dI/du -> parameter
                               sweep B -> [join] -> I, sum v^2, SSE, loss values            (the critical path: one stream,
                               dependent launches of one stream follow each other without the ~10 us cross-queue hop)
                        side 1 u-forward (interior + boundary, one launch) -> boundary residual -> cotangent A
                               -> parameter sweeps {interior/A, boundary} (one launch; sweep A also returns nabla_x u:
                               same adjoint as the helper backward)"""
        th = self.theta.data
        M = (self.method, self.H, self.K, self.m)
        e0 = self._mark()
        fused_x = self.pollution == 1.0 and not self.adjoint
        joint = G.Nb and G.same_grid           # boundary paths on the interior's time grid: one launch for both
        e_x = None
        if self._compact(G, fused_x, joint):
            return self._gen_front_compact(G)
        if not getattr(G, 'skip_v', False):
            self._launch_test_net_here(G)                        # enqueued first: its blocks must be resident before the
        with self._side(1, e0):                                  # stepper's waves spread over the CUs
            fwd = [self._job(G, 'i')] + ([self._job(G, 'b')] if joint else [])
            KN.ode_fwd_multi(fwd, G.t, th, *M, zero16=self.scal, narrow=self._narrow_ok(fwd, alone=False, forward=True),
                             prio_drop=self.prio_drop['G'])
            if G.Nb and not joint:
                fwd_b = [self._job(G, 'b')]
                KN.ode_fwd_multi(fwd_b, G.tb, th, *M, narrow=self._narrow_ok(fwd_b, alone=False, forward=True))
            self._reaction(G)
            e_f = self._mark()
            # Cotangent A (pollution + the initial-value penalty at t_0) and the boundary cotangent are residuals of what
            # the forward pass just wrote: the sweeps form them on the fly (XwOdeBwdJob.res_*) and start right behind the
            # forward pass -- the boundary sum of squares (a loss value, not an input of any sweep) runs beside them.  One
            # launch and two [L, N] buffers fewer per sub-step; the cycle does not change (1885 steps/s either way): the
            # sweeps start 26 us earlier, next to the test network, whose launch then lasts that much longer -- the first
            # phase of the sub-step is bound by SIMD time, not by this chain.
            # (pairwise group: the initial penalty is the mean over all PAIRS (u_n - h_m)^2, whose u-gradient is 2 (u_n - mean h) / N)
            res_A = dict(u=G.u, ref=G.href if G.pair_i else G.h, coef=2.0 * self.alpha / G.Nglob, base=self.pollution, first_only=True)
            res_b = dict(u=G.ub, ref=G.g, coef=2.0 * self.alpha / (G.Nbglob * G.Lb), base=0.0, first_only=False) if G.Nb else None
            e_b = None       # (the boundary sum of squares, a loss value only, is formed by the reduction at the end: _contract(with_bdry))
            # With the reference's pollution (cotangent A = ones + the initial-value term at t_0) sweep A and the helper
            # backward u.backward(ones) are the same adjoint: one launch returns the parameter gradient of A and nabla_x u.
            if not fused_x:
                with self._side(2, e_f):
                    sweep_x = [self._job(G, 'i', want_x=True)]
                    KN.ode_bwd_multi(sweep_x, G.t, th, *M, want_x=True, want_params=False, adjoint=self.adjoint,
                                     narrow=self._narrow_ok(sweep_x, alone=False, params=False))
                    e_x = self._mark()
            sweeps = [dict(self._job(G, 'i', None, G.slabA[:G.ns_u], want_x=fused_x), res=res_A)]
            if joint:
                sweeps.append(dict(self._job(G, 'b', None, G.slabA[G.ns_u:]), res=res_b))
            KN.ode_bwd_multi(sweeps, G.t, th, *M, want_x=fused_x, want_params=True, x_cot_ones=fused_x, adjoint=self.adjoint,
                             narrow=self._narrow_ok(sweeps, alone=False), prio_drop=self.prio_drop['A'])
            if G.Nb and not joint:
                sweep_b = [dict(self._job(G, 'b', None, G.slabA[G.ns_u:]), res=res_b)]
                KN.ode_bwd_multi(sweep_b, G.tb, th, *M, want_x=False, want_params=True, adjoint=self.adjoint,
                                 narrow=self._narrow_ok(sweep_b, alone=False))
            e_A = self._mark()
            # The slabs of sweeps A + boundary are summed HERE, beside the tail of sweep B, so that the update at the end of the sub-step
            # only has sweep B's slabs left to read (k_slab_sum is the A half of k_adam's own summation tree: the same bits).
            G.sumA_ready = self.early_slab_sum and self._world_of(G) is None and self.accum_u is None and self.use_streams
            e_S = None
            if G.sumA_ready:
                if getattr(G, 'sumA', None) is None:
                    G.sumA = torch.empty(self.Pu, dtype=F64, device=self.dev)
                KN.slab_sum(G.slabA, out=G.sumA)
                e_S = self._mark()
        e_v = self._mark()
        self._join(e_f)
        # cotangent B = dI/du is a pointwise product of what the two forward passes wrote: sweep B forms it on the fly too
        # (XwOdeBwdJob.res_first_only = 2) and starts right behind the test network: one launch fewer on the critical chain
        # test network -> sweep B -> Adam (0.5063 -> 0.5034 ms per generator sub-step in the same run)
        res_B = dict(u=G.u, ref=G.v, coef=G.Vol / G.Nglob / G.L * G.s3_scale, base=G.Vol / G.Nglob,
                     weak=dict(w=G.w, c=G.c, cp=G.cp, ckappa=G.ck))
        sweep_B = [dict(self._job(G, 'i', None, G.slabB), res=res_B)]
        KN.ode_bwd_multi(sweep_B, G.t, th, *M, want_x=False, want_params=True, adjoint=self.adjoint,
                         narrow=self._narrow_ok(sweep_B, alone=True))
        # the reduction needs nabla_x u (sweep A) and v, not sweep B: it runs behind sweep A on the side stream, next to
        # the tail of sweep B, instead of after it
        with self._side(3, e_A, e_v, *[e for e in (e_x, e_b) if e is not None]):   # (re-entering side 1 here crashes hipStreamEndCapture)
            self._contract(G, self.adam_u, with_bdry=True)       # -> scal[0..3], loss values
            e_C = self._mark()
        self._join(e_C, e_S)

    def _compact(self, G, fused_x, joint):
        """small groups (shards of a strong-scaling job, small problems): the sub-step is a chain of dependent launches whose
        every cross-queue dependency edge costs ~12 us -- the compact schedule below has one instead of three"""
        tiles = (G.N + 15) // 16 + ((G.Nb + 15) // 16 if G.Nb else 0)
        if not (self.compact_tiles > 0 and self.use_streams and fused_x and joint):
            return False
        # ... and ANY group whose test network is not evaluated in this sub-step (v, dv/dt, nabla_x v(t_0) reused while phi and the
        # sample are unchanged: the second generator sub-iteration of train()): nothing holds sweep B back, the chip is empty, and
        # the three sweep jobs in one launch take 146 us where sweeps A + boundary (139) and B (94) followed each other -- as branches
        # of the wide graph they land on one hardware queue behind the reduction (profiles/r06_train_timeline.txt)
        return tiles <= self.compact_tiles or bool(getattr(G, 'skip_v', False))

    def _gen_front_compact(self, G):
        """_gen_front for a group that leaves the chip mostly idle: test network (main) || forward pass (side 1), then ON THE MAIN
        STREAM one launch with all three sweep jobs (A with nabla_x u, boundary, B), the reduction and (in _gen_back) the update --
        same kernels, same arguments, same results as the wide schedule; dependent launches of one stream follow each other without
        the cross-queue hop."""
        th = self.theta.data
        M = (self.method, self.H, self.K, self.m)
        e0 = self._mark()
        if not getattr(G, 'skip_v', False):
            self._launch_test_net_here(G)
        with self._side(1, e0):
            fwd = [self._job(G, 'i'), self._job(G, 'b')]
            lone = bool(getattr(G, 'skip_v', False))
            KN.ode_fwd_multi(fwd, G.t, th, *M, zero16=self.scal, narrow=self._narrow_ok(fwd, alone=lone, forward=True),
                             prio_drop=self.prio_drop['G'])
            self._reaction(G)
            e_f = self._mark()
        self._join(e_f)
        res_A = dict(u=G.u, ref=G.href if G.pair_i else G.h, coef=2.0 * self.alpha / G.Nglob, base=self.pollution, first_only=True)
        res_b = dict(u=G.ub, ref=G.g, coef=2.0 * self.alpha / (G.Nbglob * G.Lb), base=0.0, first_only=False)
        res_B = dict(u=G.u, ref=G.v, coef=G.Vol / G.Nglob / G.L * G.s3_scale, base=G.Vol / G.Nglob,
                     weak=dict(w=G.w, c=G.c, cp=G.cp, ckappa=G.ck))
        sweeps = [dict(self._job(G, 'i', None, G.slabA[:G.ns_u], want_x=True), res=res_A),
                  dict(self._job(G, 'b', None, G.slabA[G.ns_u:]), res=res_b),
                  dict(self._job(G, 'i', None, G.slabB), res=res_B)]
        KN.ode_bwd_multi(sweeps, G.t, th, *M, want_x=True, want_params=True, x_cot_ones=True, adjoint=self.adjoint,
                         narrow=self._narrow_ok(sweeps, alone=bool(getattr(G, 'skip_v', False))))
        G.sumA_ready = False
        self._contract(G, self.adam_u, with_bdry=True)

    def begin_substep(self, which, accumulate):
        """start of a generator ('u') / discriminator ('v') sub-iteration over several groups: zero the carried gradient"""
        name, P = ('accum_u', self.Pu) if which == 'u' else ('accum_v', self.Pv)
        if which == 'u':
            self._field_touched = False
        if not accumulate:
            setattr(self, name, None)
        elif getattr(self, name) is None:
            setattr(self, name, torch.zeros(P, dtype=F64, device=self.dev))
        else:
            getattr(self, name).zero_()

    def _gen_back(self, G):
        lr, st = self.config['u_rate'], self.adam_u
        acc = self.accum_u
        world = self._world_of(G)
        lag = dict(lag=st['lag'], lag_range=self.field_range, skip=self.adam_skips_untouched and not self._field_seen(G))
        if world is not None:     # (the whole group here: loss values and counter were done by the reduction launch, _contract)
            if G.pair_i:
                KN.pair_fold(self.scal, G.Vol, G.Nglob)
            KN.losses(self.scal, G.L, G.Lb, G.Vol, G.Nglob, G.Nbglob, self.alpha, step=st['step'], init_off=G.init_off,
                      bdry_off=G.bdry_off)
        if world is None and getattr(G, 'sumA_ready', False) and acc is None:
            KN.adam(self.theta.data, None, st['m'], st['v'], st['step'], lr, gslabB=G.slabB, scal=self.scal,
                    gextraA=G.sumA, gsum_out=self.grad_u, bump_step=-1, **lag)
        elif world is None:
            KN.adam(self.theta.data, G.slabA, st['m'], st['v'], st['step'], lr, gslabB=G.slabB, scal=self.scal,
                    gextraA=acc, gsum_out=self.grad_u, bump_step=-1, **lag)
        else:
            P = self.Pu
            if acc is not None:
                self.pack_u[:P].add_(acc)
            KN.adam(self.theta.data, None, st['m'], st['v'], st['step'], lr, gextraA=self.pack_u[:P],
                    gextraB=self.pack_u[P:2 * P], scal=self.scal, gsum_out=self.grad_u, bump_step=-1, **lag)
        if acc is not None:
            acc.copy_(self.grad_u)

    def _v_fresh(self, G, store=False):
        """python-side bookkeeping (outside the captured graphs): are the test-network outputs of this group still those
        of the current phi and sample?  Sets G.skip_v for the front segment and returns the graph-key suffix."""
        now = self._v_key(G)
        G.skip_v = self.reuse_test_net and getattr(G, 'v_version', None) == now
        G.v_version = now
        if not G.skip_v:                 # this sub-step evaluates the test network: does it leave the layer inputs behind?
            G.vact_valid = G.vact is not None and (store or self.reuse_test_net)
        return ('_vcached' if G.skip_v else '') + ('_act' if getattr(G, 'vact_valid', False) else '')

    def _v_key(self, G):
        """(engine-side updates of phi, torch-side in-place writes to its parameters, resamples of the group)"""
        return (self._phi_version, sum(p._version for p in self.phi.params), G.sample_version)

    def invalidate_test_net(self):
        """phi was changed from outside the engine (optimizer_v.step(), load_state_dict, ...)"""
        self._phi_version += 1

    def _run_runner(self, fn, *args):
        from ._lib import check
        self._exchange_error = None
        rc = fn(*args)
        if getattr(self, '_exchange_error', None) is not None:      # (raised inside the host-side exchange stand-in)
            err, self._exchange_error = self._exchange_error, None
            raise err
        check(rc, fn.__name__)

    def generator_step(self, G):
        """one pass of the generator sub-step body; loss_u is left in scal[4] (device)"""
        sfx = self._v_fresh(G)
        world = self._world_of(G)
        if self._runner_ok(G):
            from ._lib import lib
            skip = self.adam_skips_untouched and not self._field_seen(G)
            G.ck = self.structure.c_kappa
            self._run_runner(lib.xw_substep_gen, self._runner_group(G), self._runner_state(), int(bool(G.skip_v)),
                             int(bool(getattr(G, 'vact_valid', False))), KN._p(self.accum_u), int(skip), KN._stream())
            return
        if world is None:
            self._run(G, 'gen' + sfx, self._gen_all)
            return
        if self.capture_exchange:
            self._run(G, 'gen_dist' + sfx, self._gen_all_dist)
            return
        self._run(G, 'gen_front' + sfx, self._gen_front_packed)   # ... -> pack_u = [sum A | sum B | scal]
        world.all_reduce(self.pack_u)                             # the ONE exchange of the generator sub-step
        self._run(G, 'gen_back', self._gen_back)

    def _gen_all_dist(self, G):
        """several GPUs, capturable exchange: front segment, the ONE all-reduce and the update as one graph"""
        self._gen_front_packed(G)
        self.world.all_reduce(self.pack_u)
        self._gen_back(G)

    def _gen_front_packed(self, G):
        """several GPUs: the front segment ends with the slab sums into the exchange buffer (same captured graph)"""
        P = self.Pu
        if G.N == 0:
            return self._gen_front_empty(G)
        self._gen_front(G)
        KN.slab_sum2(G.slabA, self.pack_u[:P], G.slabB, self.pack_u[P:2 * P])

    def _gen_front_empty(self, G):
        """_gen_front_packed on a rank whose share of the group holds no interior path (fewer paths than ranks): nothing of the
        weak form lives here -- zeros go into the exchange -- but the boundary paths the rank holds, if any, take their forward
        pass and their sweep as usual (csrc/xw_substep.hip does the same for the runner's groups)"""
        P, th = self.Pu, self.theta.data
        M = (self.method, self.H, self.K, self.m)
        self.pack_u.zero_()
        if not G.Nb:
            return
        fwd_b = [self._job(G, 'b')]
        KN.ode_fwd_multi(fwd_b, G.tb, th, *M, narrow=self._narrow_ok(fwd_b, alone=False, forward=True))
        res_b = dict(u=G.ub, ref=G.g, coef=2.0 * self.alpha / (G.Nbglob * G.Lb), base=0.0, first_only=False)
        sweep_b = [dict(self._job(G, 'b', None, G.slabA[G.ns_u:]), res=res_b)]
        KN.ode_bwd_multi(sweep_b, G.tb, th, *M, want_x=False, want_params=True, adjoint=self.adjoint,
                         narrow=self._narrow_ok(sweep_b, alone=False))
        KN.bdry_partials(G.ub, G.g, self.alpha, G.Nbglob, self.scal, G.work_b)
        KN.slab_sum(G.slabA[G.ns_u:], out=self.pack_u[:P])

    # ------------------------------------------------------------------------------------------------------------
    # discriminator sub-step (src/training.py:152-162)
    # ------------------------------------------------------------------------------------------------------------
    def _disc_front(self, G):
        """main: test network (+ its record) -> [join] -> I, sum v^2, loss values;  side 1: u-forward -> x-sweep"""
        th = self.theta.data
        M = (self.method, self.H, self.K, self.m)
        if G.N == 0:                     # an empty share of a sharded group: zeros into both exchanges
            self.scal.zero_()
            return
        e0 = self._mark()
        if not getattr(G, 'skip_v', False):
            self._launch_test_net_here(G, blocks=self.v_blocks_disc)
        with self._side(1, e0):
            # (the only sweep of this sub-step has no weight gradients: the forward stores a seventh of the record)
            fwd = [self._job(G, 'i')]
            lone = bool(getattr(G, 'skip_v', False))          # (the test network is reused: this chain has the chip to itself)
            KN.ode_fwd_multi(fwd, G.t, th, *M, zero16=self.scal, act_x_only=True, narrow=self._narrow_ok(fwd, alone=lone, forward=True),
                             prio_drop=self.prio_drop['F'])
            self._reaction(G)
            sweep_x = [self._job(G, 'i', want_x=True)]
            KN.ode_bwd_multi(sweep_x, G.t, th, *M, want_x=True, want_params=False, adjoint=self.adjoint,
                             narrow=self._narrow_ok(sweep_x, alone=lone, params=False), prio_drop=self.prio_drop['X'])
            e_x = self._mark()
        self._join(e_x)
        self._contract(G, self.adam_v)

    def _disc_mid(self, G):
        KN.disc_cotangent(G.u, G.v, G.w, G.f, G.h, G.Vol, G.Nglob, self.scal, G.vbar, c=G.c, ckappa=G.ck,
                          pollution=self.pollution, s3_scale=G.s3_scale)
        act = G.vact if getattr(G, 'vact_valid', False) else None
        if G.tpp is None:
            KN.disc_bwd(G.xvT, G.t, self.phi.data, G.vbar, self.W, self.q, gslab=G.slab_v, act=act)
        else:
            KN.disc_bwd(G.xvT_pts, None, self.phi.data, G.vbar.view(1, -1), self.W, self.q, tpp=G.tpp, gslab=G.slab_v, act=act)

    def _disc_mid_packed(self, G):
        if G.N == 0:
            self.grad_v.zero_()
            return
        self._disc_mid(G)
        KN.slab_sum(G.slab_v, out=self.grad_v)

    def _disc_back(self, G):
        lr, st = self.config['v_rate'], self.adam_v
        acc = self.accum_v
        world = self._world_of(G)
        if world is not None:
            KN.losses(self.scal, G.L, G.Lb, G.Vol, G.Nglob, G.Nbglob, self.alpha, step=st['step'])
        if world is None:
            KN.adam(self.phi.data, G.slab_v, st['m'], st['v'], st['step'], lr, gextraA=acc, gsum_out=self.grad_v,
                    bump_step=-1)
        else:
            if acc is not None:
                self.grad_v.add_(acc)
            KN.adam(self.phi.data, None, st['m'], st['v'], st['step'], lr, gextraA=self.grad_v, bump_step=-1)
        if acc is not None:
            acc.copy_(self.grad_v)

    def _reduce_sums(self, G):
        """several GPUs: the partial sums I, sum v^2, SSE (+ the two factors of a pairwise group's d(phi)/dt term, folded
        into I once they are global)"""
        self.world.all_reduce(self.scal[0:9])
        if G.pair_i:
            KN.pair_fold(self.scal, G.Vol, G.Nglob)

    def _disc_all(self, G):
        self._disc_front(G)
        self._disc_mid(G)
        self._disc_back(G)

    def _disc_all_dist(self, G):
        """several GPUs, capturable exchanges: the whole discriminator sub-step with its two all-reduces as one graph"""
        self._disc_front(G)
        self._reduce_sums(G)
        self._disc_mid_packed(G)
        self.world.all_reduce(self.grad_v)
        self._disc_back(G)

    def discriminator_step(self, G):
        """one pass of the discriminator sub-step body; loss_v is left in scal[5] (device)"""
        sfx = self._v_fresh(G, store=True)
        self._phi_version += 1                                    # phi changes at the end of this sub-step
        world = self._world_of(G)
        if self._runner_ok(G):
            from ._lib import lib
            G.ck = self.structure.c_kappa
            self._run_runner(lib.xw_substep_disc, self._runner_group(G), self._runner_state(), int(bool(G.skip_v)),
                             int(bool(getattr(G, 'vact_valid', False))), KN._p(self.accum_v), KN._stream())
            return
        if world is None:
            self._run(G, 'disc' + sfx, self._disc_all)
            return
        if self.capture_exchange:
            self._run(G, 'disc_dist' + sfx, self._disc_all_dist)
            return
        self._run(G, 'disc_front' + sfx, self._disc_front)
        self._reduce_sums(G)                                      # I and sum v^2 must be global before the cotangent
        self._run(G, 'disc_mid' + ('_act' if getattr(G, 'vact_valid', False) else ''), self._disc_mid_packed)
        world.all_reduce(self.grad_v)
        self._run(G, 'disc_back', self._disc_back)

    # ------------------------------------------------------------------------------------------------------------
    def _run(self, G, key, fn, scratch=False):
        """execute fn(G) eagerly, or capture it once into a HIP graph (per group and segment) and replay it.
        scratch: fn allocates while it runs and leaves nothing behind in what it allocated (True / 1: scratch pool 0, 2: pool 1)"""
        # (a pairwise group carries per-sample host constants -- the variance offsets -- into its launches: never captured)
        capturable = (self.use_graphs and self.accum_u is None and self.accum_v is None and getattr(G, 'persistent', True)
                      and not (G.pair_i or G.pair_b))
        g = G.graphs.get(key) if capturable else False
        if g is False:                            # not capturable, or capture of THIS segment was refused before
            if not getattr(G, 'persistent', True) and self.use_streams:
                # a group of a list domain (new shapes every sample: eager launches, issued faster than the small kernels
                # run out): one stream -- the side-stream contexts and events cost more host time than the overlap returns
                # (cone outer iteration 28.5 -> 25.4 ms, hourglass 57.5 -> 52.4 ms)
                self.use_streams = False
                # (operand validation: every launch of the first `eager_checked` eager segments -- a few outer iterations --
                #  then the engine trusts its own buffers, kernels.TRUSTED; XW_ALWAYS_CHECK=1 keeps validating)
                self._eager_seen = getattr(self, '_eager_seen', 0) + 1
                KN.TRUSTED = self._eager_seen > self.eager_checked
                try:
                    fn(G)
                finally:
                    self.use_streams = True
                    KN.TRUSTED = False
                return
            fn(G)
            return
        if g is None:
            ran_eager = self.structure.c_kappa is None
            if ran_eager:
                fn(G)                             # a black-box c(u, t, x): one eager pass first (allocations, lazy init of its ops)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            try:
                # thread_local: other threads (the RCCL watchdog polls events) must not invalidate the capture
                # (HIP_HOST_LOCK: no page-locked allocation of the sampling helper thread during a capture or a launch)
                with HIP_HOST_LOCK, torch.cuda.graph(g, pool=_scratch_pool(int(scratch) - 1) if scratch else None, stream=self._capture_stream(),
                                                     capture_error_mode='thread_local'):
                    fn(G)
            except Exception as exc:
                # Capture refused (typically a user callable that syncs with the host or builds CPU tensors): THIS segment
                # of THIS group runs eagerly from now on -- several times slower, so say it loudly, once per segment, and
                # leave every other segment captured (XW_STRICT_GRAPHS=1 turns the cliff into an error)
                if self.options.strict_graphs:
                    raise
                import warnings
                warnings.warn('HIP graph capture of sub-step segment %r failed (%s: %s); this segment now runs as eager kernel launches '
                              '(expect a several times lower step rate)' % (key, type(exc).__name__, exc), RuntimeWarning, stacklevel=2)
                G.graphs[key] = False
                self.eager_segments = getattr(self, 'eager_segments', 0) + 1
                torch.cuda.synchronize()
                if not ran_eager:                 # (with a black-box c the eager pass above already WAS this call's step:
                    fn(G)                         #  running fn again would apply a second optimiser update)
                return
            G.graphs[key] = g
            _KEPT_GRAPHS.append(g)                # (never destroyed: see _KEPT_GRAPHS)
            if len(_KEPT_GRAPHS) in (256, 1024, 4096):
                import warnings
                # a process that builds many solvers (hyper-parameter sweeps) or whose group shapes keep changing: every graph
                # and its private memory pool stay until exit
                warnings.warn('%d captured HIP graphs are being kept alive (destroying one faults a later launch on this runtime, '
                              'engine._KEPT_GRAPHS): a long-lived process that keeps building solvers should run them with '
                              'XW_GRAPHS=0 or in child processes' % len(_KEPT_GRAPHS), RuntimeWarning, stacklevel=3)
            if ran_eager:
                return                            # (the eager pass above was this call's step; the capture only recorded)
        with HIP_HOST_LOCK:
            g.replay()

    def _capture_stream(self):
        return self._cap

    def loss_u(self):
        return self.scal[4]

    def loss_v(self):
        return self.scal[5]

    def predict_group(self, G):
        """u_theta on the interior paths of a loaded group as the module returns it, [N, L, 1] -- no path tensor, no callables:
        the group already holds the transposed points, the grid and the start values"""
        u, _ = KN.ode_fwd(G.xT, G.t, G.start, self.theta.data, self.method, self.H, self.K, self.m, want_Y=False)
        return u.t().unsqueeze(2)

    def predict(self, X):
        """u_theta on a group [N, L, d+1] -> [L, N] (diagnostics; no checkpoints kept)"""
        Xd = X.detach()
        starts_T0 = float(Xd[0, 0, 0]) == self.setup['T0']
        s = self.funcs['h'](Xd[:, 0, :]) if starts_T0 else self.funcs['g'](Xd[:, 0, :].unsqueeze(1)).reshape(-1)
        u, _ = KN.ode_fwd(Xd[:, 0, 1:].to(self.dev).to(F64).t().contiguous(), Xd[0, :, 0].to(self.dev).to(F64).contiguous(),
                          s.detach().to(self.dev).to(F64).reshape(-1).contiguous(), self.theta.data, self.method,
                          self.H, self.K, self.m, want_Y=False)
        return u
