"""Multi-GPU: one process per GPU, Monte-Carlo paths sharded contiguously along the path axis, parameters replicated.

The reference's only parallelism is nn.DataParallel around both nets (src/training.py:93-97): per forward it
broadcasts the parameters, scatters the [N, L, d+1] batch along dim 0 and gathers the outputs; per backward it
reduce-adds the gradients.  Here nothing but two small buffers ever crosses xGMI per optimiser sub-step:

  generator sub-step: ONE all-reduce of [ J^T ubarA (P_u) | J^T ubarB (P_u) | I, sum v^2, SSE_init, SSE_bdry ]
     (2 P_u + 16 doubles = 26 KB at d = 20).  The loss is log(I^2) - log(V S / P) + alpha (...), NOT a sum over paths,
     but its theta-gradient is  J^T ubarA + (2/I) J^T ubarB  with two cotangent bases that need no global scalar
     (xw_gen_cotangents); the 2/I factor is applied after the exchange, inside the fused Adam kernel.
  discriminator sub-step: TWO all-reduces -- (I, sum v^2) (32 bytes) before the cotangent
     vbar = w - (2/I) dI/dv + 2 v / S can be formed, then the packed gradient (P_v doubles, 30 KB at d = 20).  Folding
     them into one would need three backward passes through the test network (the dominant kernel) instead of one.
Every rank applies the identical fused Adam update, so parameters stay bit-identical without broadcasts.  All
messages are latency-bound (RCCL LL protocol on the fully connected xGMI mesh).  The 1/N, 1/(N L), 1/(N_b L) factors
use GLOBAL counts on every rank.
"""
import ctypes
import os

import torch
import torch.distributed as dist


class World:
    """The ranks of one job.  On the `nccl` (= RCCL) backend the exchanges go through the library's own entry point
    xw_allreduce (include/xnwan.h) on a communicator created here: the call only enqueues on torch's current stream, so
    the engine captures a sub-step TOGETHER with its exchange into one HIP graph (Engine.capture_exchange).
    torch.distributed remains the rendezvous (it carries the 128-byte RCCL id to the ranks) and the `gloo` rehearsal path."""

    def __init__(self, group=None, native=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.comm = None
        from .options import EngineOptions
        self.replicate_below = int(EngineOptions.from_env().replicate_below)
        if native is None:
            from .options import EngineOptions
            native = dist.get_backend(group) == 'nccl' and EngineOptions.from_env().native_allreduce
        if native:
            self._init_native()

    def _init_native(self):
        from ._lib import lib
        import warnings
        flag_dev = 'cuda' if dist.get_backend(self.group) == 'nccl' else 'cpu'

        def all_agree(ok_here):
            ok = torch.tensor([1 if ok_here else 0], device=flag_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            return int(ok[0]) == 1
        # ncclCommInitRank is COLLECTIVE: a rank that cannot bind RCCL would return from xw_comm_init at once and leave
        # the others blocked in the bootstrap.  So the ranks first agree (non-collective probe + one MIN all-reduce through
        # torch.distributed) that every one of them has the symbols, and only then enter it.
        have = lib.xw_comm_available() == 0
        if not all_agree(have):
            warnings.warn('RCCL could not be bound on a rank (%s here): the exchanges go through torch.distributed between graph '
                          'segments instead of RCCL calls inside the captured sub-steps' % ('available' if have else 'NOT available'),
                          RuntimeWarning, stacklevel=3)
            return
        ident = ctypes.create_string_buffer(128)
        have_id = self.rank == 0 and lib.xw_comm_unique_id(ident) == 0
        box = [ident.raw if have_id else None]
        dist.broadcast_object_list(box, src=0, group=self.group)
        if box[0] is None:                # rank 0 could not create the id: every rank sees None and falls back together
            warnings.warn('xw_comm_unique_id failed on rank 0: exchanges through torch.distributed', RuntimeWarning, stacklevel=3)
            return
        comm = ctypes.c_void_p()
        rc = lib.xw_comm_init(box[0], self.size, self.rank, ctypes.byref(comm))
        # every rank must end up on the same path: agree on the outcome (a rank whose communicator failed AFTER joining the
        # bootstrap sends all of them to torch.distributed's all-reduce between graph segments, loudly)
        if all_agree(rc == 0):
            self.comm = comm
            return
        if rc == 0:
            lib.xw_comm_destroy(comm)
        warnings.warn('xw_comm_init failed on a rank (status %d here): the exchanges go through torch.distributed between '
                      'graph segments instead of RCCL calls inside the captured sub-steps' % rc, RuntimeWarning, stacklevel=3)

    @property
    def capturable(self):
        """exchanges may be recorded into a HIP graph (device-side call on the current stream, no host staging)"""
        return self.comm is not None

    def all_reduce(self, t):
        if self.comm is not None:
            from ._lib import lib, check
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
                raise RuntimeError('xw_allreduce sums contiguous float64 device buffers')
            check(lib.xw_allreduce(t.data_ptr(), t.numel(), self.comm, torch.cuda.current_stream().cuda_stream), 'xw_allreduce')
            return t
        if t.is_cuda and dist.get_backend(self.group) == 'gloo':
            # rehearsal mode (several ranks sharing one GPU, or no RCCL): stage through the host
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def bounds(self, n):
        """contiguous, balanced split of n paths: rank r owns [lo, hi)"""
        base, rem = divmod(n, self.size)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def assert_in_step(self, *blobs):
        """Raise on every rank if the ranks' copies of the given float64 parameter blobs are not BIT-identical.  Parameters are
        replicated and never broadcast (every rank applies the same fused Adam to the same all-reduced gradient; groups below
        `replicate_below` are computed whole on every rank), so a divergence can only be noticed by looking.  Per blob a 63-bit
        position-weighted hash of the bit patterns (exact integer arithmetic, wrapping), cut into three 21-bit pieces p; the ranks
        exchange sum(p) and sum(p^2) -- both exact in float64 -- and  size * sum(p^2) == sum(p)^2  holds iff every rank sent the
        same p (Cauchy-Schwarz; exact up to 32 ranks: (32 x 2^21)^2 < 2^53).  One small all-reduce and one read-back: called once per
        train(), not per sub-step."""
        pieces = []
        for b in blobs:
            bits = b.detach().reshape(-1).to(torch.float64).contiguous().view(torch.int64)
            w = torch.arange(1, 2 * bits.numel(), 2, dtype=torch.int64, device=bits.device)         # odd multipliers: a permutation changes the hash
            h = int((bits * w).sum().item()) & ((1 << 63) - 1)                                      # (int64 products and sums wrap: exact mod 2^64)
            pieces += [float(h & 0x1FFFFF), float((h >> 21) & 0x1FFFFF), float((h >> 42) & 0x1FFFFF)]
        dev = blobs[0].device
        mine = torch.tensor(pieces + [p_ * p_ for p_ in pieces], dtype=torch.float64, device=dev)
        tot = mine.clone()
        self.all_reduce(tot)
        n = len(pieces)
        bad = bool((self.size * tot[n:] != tot[:n] * tot[:n]).any().item())                         # (every rank sees the same verdict)
        if bad:
            raise RuntimeError('rank %d of %d: the replicated parameters have drifted apart across the ranks (hash pieces here %r)'
                               % (self.rank, self.size, [int(p_) for p_ in pieces]))

    def close(self):
        if self.comm is not None:
            from ._lib import lib
            lib.xw_comm_destroy(self.comm)
            self.comm = None

    def replicated(self, n, nb):
        """A group this small is not sharded at all: every rank computes ALL of it, with the single-process arithmetic and
        no exchange (identical inputs, deterministic kernels: the replicas stay bit-identical as they do through the
        replicated Adam update).  A 16-path tile is the stepper's unit of work and the groups in question are latency-
        bound, so a shard of less than a tile per rank does not shorten any launch -- it only adds the two or three
        exchanges of a sub-step.  The time-varying ball domains produce such groups in every sample (config 5, hourglass:
        interior groups of 5, 3, 3, 4, 2 paths next to one of 4991).  XW_REPLICATE_BELOW paths per rank (default 16; 0: never --
        every group is sharded, ranks whose share is empty stay in step through shard_group / the engine's empty shares)."""
        lim = self.replicate_below * self.size
        return n < lim and nb < lim

    replicate_below = 16          # (instance attribute, set in __init__ from options.EngineOptions.replicate_below)

    def shard_group(self, du, dv, bd):
        """this rank's slice of a group (every rank sampled the same global group from the same seed).  TOTAL: a group with
        fewer paths than ranks leaves some ranks with an empty slice (shape [0, L, d+1]) of the interior sample, of the boundary
        sample or of both -- such a rank launches nothing for what it does not hold but joins every exchange of the group's
        sub-step with zeros and applies the same update (engine.Engine; the reference's nn.DataParallel scatters a batch of any
        size, src/training.py:93-97)."""
        n, nb = du.shape[0], bd.shape[0]
        lo, hi = self.bounds(n)
        blo, bhi = self.bounds(nb)
        us = du[lo:hi]
        return us, (us if dv is du else dv[lo:hi]), bd[blo:bhi], n, nb


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns (World or None, local device index)"""
    size = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if size == 1:
        return None, local
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = os.environ.get('XW_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()      # rehearsal: more ranks than GPUs share devices (gloo only)
    if backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device('cuda', local))
    else:
        dist.init_process_group(backend)
    return World(), local
