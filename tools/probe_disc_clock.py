"""in-kernel shader clock of k_disc_fwd (diagnostic library built with -DXW_CLOCK_PROBE): d(s_memtime) / d(s_memrealtime) x 100 MHz
over every wave's tile loop, after ~2 s of back-to-back launches; on random and on all-zero operands.
usage: XW_LIBRARY=_var/libxnwan_clock.so python tools/probe_disc_clock.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
d, N, L, W, q = 20, 4096, 32, 50, 9
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
Pv = _lib.lib.xw_phi_size(d, W)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
lib = _lib.lib
lib.xw_debug_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, scale in (('random', 1.0), ('zeros', 0.0), ('random', 1.0)):
    ph = (scale * 0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
    xT = (scale * torch.rand(d, N, generator=g, dtype=torch.float64)).to(dev)
    for blocks in (0, 256, 352):
        fn = lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N, max_blocks=blocks)  # noqa: E731
        t0 = time.time()
        while time.time() - t0 < 2.0:
            for _ in range(50): fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        nw = 4 * (blocks or 512)
        buf = np.zeros(2 * nw, dtype=np.uint64)
        assert lib.xw_debug_clock(buf.ctypes.data, nw) == 0
        ck, rt = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
        mhz = ck / rt * 100.0
        print('%-7s blocks %3d: %.1f us per launch; in-kernel clock median %.0f MHz (min %.0f max %.0f); wave loop median %.0f kclk = %.1f us'
              % (name, blocks or 512, e0.elapsed_time(e1) / 50 * 1e3, np.median(mhz), mhz.min(), mhz.max(), np.median(ck) / 1e3, np.median(rt) / 100))
