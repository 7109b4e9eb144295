"""Shader clock and power while the test-network forward runs back to back (is the FP64 matrix peak of 78.6 TFLOP/s
= 2.4 GHz reachable in sustained operation?).  Samples rocm-smi from a side thread; prints idle and loaded readings
and the kernel's average duration over each second."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib

N, L, d, W, q = 4096, 32, 20, 50, 9
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev); t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=20).stdout
        return ' '.join(out.split())[:600]
    except Exception as exc:  # noqa: BLE001
        return 'rocm-smi failed: %r' % (exc,)


print('idle  :', smi(), flush=True)
stop = False
samples = []


def sampler():
    while not stop:
        samples.append(smi())
        time.sleep(0.5)


th = threading.Thread(target=sampler); th.start()
for sec in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 2000
    e0.record()
    for _ in range(n):
        KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt)
    e1.record(); torch.cuda.synchronize()
    print('window %d: %.1f us per launch' % (sec, 1e3 * e0.elapsed_time(e1) / n), flush=True)
stop = True; th.join()
for s_ in samples[:8]:
    print('loaded:', s_)
