"""latency of the exchange step on ONE rank (RCCL communicator of size 1 behind xw_allreduce): eager calls on the stream and
replays of a captured graph that holds 20 of them, for the three message sizes of the sub-steps (generator: 2 P_u + 16
doubles, discriminator: 9 doubles and P_v doubles).  What a sub-step pays for HAVING an exchange node in its graph; the
wire time of 2..8 ranks on xGMI comes on top (profiles/r03_other_configs_1gpu.md uses both)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch
import torch.distributed as dist
from xnode_wan_pde_solver_amd import dist as xdist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
world = xdist.World()
assert world.capturable
for n, what in ((2 * 1551 + 16, 'generator pack (d=20)'), (9, 'discriminator sums'), (3651, 'discriminator gradient (d=20)')):
    buf = torch.ones(n, dtype=torch.float64, device='cuda')
    for _ in range(10): world.all_reduce(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): world.all_reduce(buf)
    e1.record(); torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / 200 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        for _ in range(20): world.all_reduce(buf)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print('%-32s %5d doubles: eager %.1f us per call, inside a captured graph %.1f us per call' % (what, n, eager, e0.elapsed_time(e1) / 400 * 1e3))
world.close(); dist.destroy_process_group()
