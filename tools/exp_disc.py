"""A/B of a library variant (XW_LIBRARY=_var/libxnwan_NAME.so): the test-network kernels alone at the headline size --
forward on the whole chip, at the generator sub-step's cap (11/16 of the slots), with the record at the discriminator
sub-step's cap, the reverse from the record -- output checksums, then generator / discriminator / cycle times through the
engine.   usage: [XW_LIBRARY=...] python tools/exp_disc.py [d N L]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
d, N, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 4096, 32)
W, q = 50, 9
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
Pv = _lib.lib.xw_phi_size(d, W)
ph = (0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
slv = torch.empty(KN.disc_bwd_slabs(N, L), Pv, dtype=torch.float64, device=dev)
vact = torch.empty(KN.disc_act_rows(W, q), KN.disc_act_cols(L * N), dtype=torch.float64, device=dev)
cus = torch.cuda.get_device_properties(dev).multi_processor_count


def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flop = 2.0 * (2 * N * L + N) * ((d + 1) * W + q * W * W + W)
fw = lambda mb, act=None: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt, gxv=gxv, gtv=gtv, ngrad=N, max_blocks=mb, act=act)  # noqa: E731
res = {}
res['fwd_full'] = timeit(lambda: fw(0))
res['fwd_11_16'] = timeit(lambda: fw(11 * 2 * cus // 16))
res['fwd_1wave_per_simd'] = timeit(lambda: fw(cus))
res['fwd_rec_7_8'] = timeit(lambda: fw(7 * 2 * cus // 8, vact))
res['rec_bwd'] = timeit(lambda: KN.disc_bwd(xT, t, ph, vbar, W, q, gslab=slv, act=vact))
print(os.environ.get('XW_LIBRARY', 'default'), ' '.join('%s %.1f us' % kv for kv in res.items()),
      'solo frac %.3f' % (flop / (res['fwd_full'] * 1e-6) / 78.6e12),
      'checksum %.15e %.15e %.15e %.15e' % (float(v.sum()), float(vt.sum()), float(gxv.sum()), float(slv.sum())))
if len(sys.argv) <= 3 or os.environ.get('XW_EXP_CYCLE'):
    import subprocess
    subprocess.run([sys.executable, 'tools/step_times.py', '90'])
