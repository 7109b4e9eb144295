"""where the waves of k_disc_rec spend their clocks (diagnostic library built with -DXW_REC_PROBE: s_memtime stamps between the
phases of a layer, summed per wave over the launch).  usage: python tools/probe_rec_phases.py _var/libxnwan_recprobe.so
(the stamps drain the LDS queue, so the phases are a little slower than in the shipped kernel; the split is what counts)"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from xnode_wan_pde_solver_amd import _lib
d, N, L, W, q = 20, 4096, 32, 50, 9
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
Pv = _lib.lib.xw_phi_size(d, W)
ph = (0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
vact = torch.empty((q + 1) * W, L * N, dtype=torch.float64, device=dev)
h = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for f in ('xw_disc_fwd', 'xw_disc_bwd'):
    getattr(h, f).argtypes = _lib.SIGNATURES[f]
h.xw_debug_rec_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
st = torch.cuda.current_stream().cuda_stream
assert h.xw_disc_fwd(xT.data_ptr(), t.data_ptr(), 0, ph.data_ptr(), N, L, d, W, q, v.data_ptr(), vt.data_ptr(), 0, 0, 0, 512, vact.data_ptr(), st) == 0
nslab = h.xw_disc_bwd_slabs(N, L)
slab = torch.empty(nslab, Pv, dtype=torch.float64, device=dev)
fn = lambda: h.xw_disc_bwd(xT.data_ptr(), t.data_ptr(), 0, ph.data_ptr(), vbar.data_ptr(), N, L, d, W, q, vact.data_ptr(), slab.data_ptr(), st)  # noqa: E731
for _ in range(20): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
buf = np.zeros(12 * 2048, dtype=np.uint64)
assert h.xw_debug_rec_clock(buf.ctypes.data, 2048) == 0
full = buf.reshape(2048, 12).astype(np.float64)
c = full[:, :8]
names = ['select+mask+transpose+loads', 'chain', 'tile(2,2)', 'barrier 1', 'outer products', 'barrier 2', 'tile prologue+output layer', 'input layer']
tot = c.sum(1)
print('%.1f us per launch; per wave %.0f kclk (min %.0f max %.0f)' % (e0.elapsed_time(e1) / 20 * 1e3, np.median(tot) / 1e3, tot.min() / 1e3, tot.max() / 1e3))
print('  kernel per wave %.0f kclk = %.1f us (%.0f MHz); prologue %.1f kclk, epilogue %.1f kclk' % (np.median(full[:, 10]) / 1e3, np.median(full[:, 11]) / 100,
      np.median(full[:, 10] / full[:, 11]) * 100, np.median(full[:, 8]) / 1e3, np.median(full[:, 9]) / 1e3))
for i, n_ in enumerate(names):
    print('  %-30s %7.1f kclk  %5.1f %%   (per wave role: %s)' % (n_, np.median(c[:, i]) / 1e3, 100 * c[:, i].sum() / tot.sum(),
          ' '.join('%.1f' % (np.median(c[w::4, i]) / 1e3) for w in range(4))))
