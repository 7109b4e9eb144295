"""What a handle on the sampling process costs (sampler_proc.py: ONE child per process, forked at the first use): the first
handle forks and page-locks the slots, later ones reuse both -- fresh, after 20 GB of device tensors, after 2 GB of page-locked
memory, after 1 GB of pageable memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xnode_wan_pde_solver_amd import sampling, sampler_proc
setup = dict(shape_param=1.0, dim=10, T0=0.0, T=1.0, N_t=20)
def once(tag):
    t0 = time.perf_counter()
    sp = sampler_proc.SamplerProcess(sampling.NSphere_THourglass, setup, 8192, 8192)
    t1 = time.perf_counter()
    sp.begin(); d, s = sp.first(); sp.shutdown()
    t2 = time.perf_counter()
    sp.close()
    t3 = time.perf_counter()
    print('%-40s start %.3f s (%s), first sample %.3f s, close %.3f s' % (tag, t1 - t0, 'pid %d' % sp.proc.pid, t2 - t1, t3 - t2), flush=True)
torch.cuda.init(); torch.zeros(1, device='cuda')
once('fresh process')
once('again')
keep = [torch.empty(1 << 30, dtype=torch.uint8, device='cuda') for _ in range(20)]
once('with 20 GB of device tensors')
pin = [torch.empty(1 << 28, dtype=torch.uint8).pin_memory() for _ in range(8)]
once('and 2 GB of page-locked host memory')
host = torch.zeros(1 << 30, dtype=torch.uint8)
once('and 1 GB of touched pageable memory')
