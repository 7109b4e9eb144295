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
import os

import torch
import torch.distributed as dist


class World:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def all_reduce(self, t):
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

    def shard_group(self, du, dv, bd):
        """this rank's slice of a group (every rank sampled the same global group from the same seed)"""
        n, nb = du.shape[0], bd.shape[0]
        lo, hi = self.bounds(n)
        blo, bhi = self.bounds(nb)
        if hi - lo == 0 or bhi - blo == 0:
            raise RuntimeError('group of %d/%d paths is too small for %d ranks' % (n, nb, self.size))
        return du[lo:hi], dv[lo:hi], bd[blo:bhi], n, nb


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
