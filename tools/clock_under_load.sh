#!/bin/bash
# shader clock reported by rocm-smi while (a) nothing runs, (b) one 256-tile sweep job loops, (c) three jobs loop, (d) the test network loops
smi() { /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; }
echo "idle: $(smi)"
for mode in 1 3 v; do
python - $mode <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from xnode_wan_pde_solver_amd import kernels as KN, _lib
mode = sys.argv[1]
L, d, H, K, m = 32, 20, 20, 10, 8
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
t = torch.linspace(0, 1, L, dtype=torch.float64).to(dev)
if mode == 'v':
    N, W, q = 4096, 50, 9
    ph = (0.2 * torch.randn(_lib.lib.xw_phi_size(d, W), generator=g, dtype=torch.float64)).to(dev)
    xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
    v = torch.empty(L, N, dtype=torch.float64, device=dev); vt = torch.empty_like(v)
    fn = lambda: KN.disc_fwd(xT, t, ph, W, q, v=v, vt=vt)
else:
    th = (0.3 * torch.randn(_lib.lib.xw_theta_size(d, H, K), generator=g, dtype=torch.float64)).to(dev)
    M = (1, H, K, m)
    def mkjob(N):
        xT = torch.rand(d, N, generator=g, dtype=torch.float64).to(dev)
        start = torch.randn(N, generator=g, dtype=torch.float64).to(dev); ubar = torch.randn(L, N, generator=g, dtype=torch.float64).to(dev)
        u = torch.empty(L, N, dtype=torch.float64, device=dev); Y = torch.empty(L, H, N, dtype=torch.float64, device=dev)
        act = torch.empty(L - 1, KN.ode_act_rows(1, H, K, m), KN.ode_act_cols(N), dtype=torch.float64, device=dev)
        slab = torch.empty(KN.ode_bwd_slabs(N), th.numel(), dtype=torch.float64, device=dev)
        job = dict(xT=xT, start=start, u=u, Y=Y, act=act)
        KN.ode_fwd_multi([job], t, th, *M)
        return dict(job, ubar=ubar, gslab=slab)
    jobs = [mkjob(4096) for _ in range(int(mode))]
    fn = lambda: KN.ode_bwd_multi(jobs, t, th, *M, want_x=False, want_params=True)
t0 = time.time()
while time.time() - t0 < 6.0:
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
PY
pid=$!
sleep 4.5
echo "mode $mode: $(smi)"
sleep 0.5
echo "mode $mode: $(smi)"
wait $pid
done
