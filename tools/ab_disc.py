"""interleaved A/B of the test-network forward across library variants IN ONE PROCESS (same device, same clocks):
    python tools/ab_disc.py _var/libxnwan_base.so _var/libxnwan_pref.so ...      [env AB_ROUNDS=7 AB_D=20 AB_N=4096 AB_L=32]
every round times each variant at each grid cap (whole chip 512, generator cap 352, one wave per SIMD 256, with the record
at 448); prints median and min per (variant, cap) and the output checksums."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from xnode_wan_pde_solver_amd import _lib
libs = sys.argv[1:]
d, N, L = int(os.environ.get('AB_D', 20)), int(os.environ.get('AB_N', 4096)), int(os.environ.get('AB_L', 32))
W, q = 50, 9
rounds = int(os.environ.get('AB_ROUNDS', 7))
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
Pv = _lib.lib.xw_phi_size(d, W)
ph = (0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
gxv = torch.empty(d, N, dtype=torch.float64, device=dev); gtv = torch.empty(N, dtype=torch.float64, device=dev)
rows = (q + 1) * W
vact = torch.empty(rows, (L * N + 15) // 16 * 16, dtype=torch.float64, device=dev)
H = []
for p in libs:
    h = ctypes.CDLL(os.path.abspath(p))
    h.xw_disc_fwd.argtypes = _lib.SIGNATURES['xw_disc_fwd']
    h.xw_disc_fwd.restype = ctypes.c_int
    H.append(h)
st = torch.cuda.current_stream().cuda_stream


def call(h, blocks, act, grad=True):
    rc = h.xw_disc_fwd(xT.data_ptr(), t.data_ptr(), 0, ph.data_ptr(), N, L, d, W, q, v.data_ptr(), vt.data_ptr(),
                       gxv.data_ptr() if grad else 0, gtv.data_ptr() if grad else 0, N if grad else 0, blocks, act, st)
    assert rc == 0, rc


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = [('full512', 512, 0, True), ('cap384', 384, 0, True), ('cap416', 416, 0, True), ('cap352', 352, 0, True), ('one256', 256, 0, True), ('rec448', 448, vact.data_ptr(), True),
         ('nograd512', 512, 0, False), ('full768', 768, 0, True), ('full640', 640, 0, True)]
res = {(i, c[0]): [] for i in range(len(H)) for c in cases}
for r in range(rounds):
    for i, h in enumerate(H):
        for name, blocks, act, grad in cases:
            res[(i, name)].append(timeit(lambda: call(h, blocks, act, grad)))
for i, p in enumerate(libs):
    call(H[i], 512, 0)
    torch.cuda.synchronize()
    print('%-28s' % os.path.basename(p), '  '.join('%s %.1f/%.1f' % (c[0], np.median(res[(i, c[0])]), np.min(res[(i, c[0])])) for c in cases),
          ' sums %.15e %.15e %.15e' % (float(v.sum()), float(vt.sum()), float(gxv.sum())))
