"""interleaved A/B of the test network's parameter gradient from the record (k_disc_rec) across library variants IN ONE
PROCESS:   python tools/ab_rec.py _var/libxnwan_base.so _var/libxnwan_x.so ...   [env AB_ROUNDS=7 AB_D=20 AB_N=4096 AB_L=32]
prints median / min microseconds per variant and the maximal deviation of its gradient from the first variant's"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from xnode_wan_pde_solver_amd import _lib
libs = sys.argv[1:]
d, N, L = int(os.environ.get('AB_D', 20)), int(os.environ.get('AB_N', 4096)), int(os.environ.get('AB_L', 32))
W, q = 50, int(os.environ.get('AB_Q', 9))
rounds = int(os.environ.get('AB_ROUNDS', 7))
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
Pv = _lib.lib.xw_phi_size(d, W)
ph = (0.2 * torch.randn(Pv, generator=g, dtype=torch.float64)).to(dev)
xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
vbar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
rows = (q + 1) * W
vact = torch.empty(rows, (L * N + 15) // 16 * 16, dtype=torch.float64, device=dev)
nslab = _lib.lib.xw_disc_bwd_slabs(N, L)
H = []
for p in libs:
    h = ctypes.CDLL(os.path.abspath(p))
    for f in ('xw_disc_fwd', 'xw_disc_bwd'):
        getattr(h, f).argtypes = _lib.SIGNATURES[f]
        getattr(h, f).restype = ctypes.c_int
    H.append(h)
st = torch.cuda.current_stream().cuda_stream
rc = H[0].xw_disc_fwd(xT.data_ptr(), t.data_ptr(), 0, ph.data_ptr(), N, L, d, W, q, v.data_ptr(), vt.data_ptr(), 0, 0, 0, 512, vact.data_ptr(), st)
assert rc == 0
slabs = [torch.full((nslab, Pv), float('nan'), dtype=torch.float64, device=dev) for _ in H]


def call(i):
    rc = H[i].xw_disc_bwd(xT.data_ptr(), t.data_ptr(), 0, ph.data_ptr(), vbar.data_ptr(), N, L, d, W, q, vact.data_ptr(), slabs[i].data_ptr(), st)
    assert rc == 0, rc


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = [[] for _ in H]
for r in range(rounds):
    for i in range(len(H)):
        res[i].append(timeit(lambda: call(i)))
ref = slabs[0].sum(0)
for i, p in enumerate(libs):
    gi = slabs[i].sum(0)
    print('%-28s %.1f / %.1f us   max |dg| vs first %.3e (scale %.3e)  finite %s' % (os.path.basename(p), np.median(res[i]), np.min(res[i]),
          float((gi - ref).abs().max()), float(ref.abs().max()), bool(torch.isfinite(slabs[i]).all())))
