import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from xnode_wan_pde_solver_amd import _lib
libs = sys.argv[1:]
d, N, W, q = 20, 4096, 50, 9
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
Pv = _lib.lib.xw_phi_size(d, W)
ph = (0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
st = torch.cuda.current_stream().cuda_stream
for p in libs:
    h = ctypes.CDLL(os.path.abspath(p))
    h.xw_disc_fwd.argtypes = _lib.SIGNATURES['xw_disc_fwd']
    for L in (1, 2, 4):
        t = torch.linspace(0, 1, max(L, 2), dtype=torch.float64)[:L].contiguous().to(dev)
        v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
        gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
        out = []
        for grad in (False, True):
            def call():
                rc = h.xw_disc_fwd(xT.data_ptr(), t.data_ptr(), 0, ph.data_ptr(), N, L, d, W, q, v.data_ptr(), vt.data_ptr(),
                                   gxv.data_ptr() if grad else 0, gtv.data_ptr() if grad else 0, N if grad else 0, 512, 0, st)
                assert rc == 0
            for _ in range(5): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): call()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 50 * 1e3)
        print('%-24s L=%d (%d tiles): no grad %.1f us, with grad %.1f us' % (os.path.basename(p), L, L * N // 16, out[0], out[1]))
