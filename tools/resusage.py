#!/usr/bin/env python3
"""Print a table of per-kernel register / LDS / scratch usage (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
srcs = sys.argv[1:] or ['xw_ode.hip', 'xw_disc.hip', 'xw_weak.hip']
for f in srcs:
    out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I../../include',
                          *(['-mllvm', '-amdgpu-mfma-vgpr-form=1'] if f == 'xw_ode.hip' else []), '-Rpass-analysis=kernel-resource-usage', '-c', f, '-o', '/dev/null'], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r'remark:\s+(.*?): (.*?) \[-Rpass', line)
        if not m: continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == 'Function Name':
            cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
        cur[k] = v
        if k.startswith('LDS Size'):
            n = re.sub(r'\(anonymous namespace\)::', '', cur['name'])
            n = re.sub(r'\(.*', '', n)
            print('%-46s VGPR %4s AGPR %4s SGPR %4s spillV %4s scratch %5s occ %s LDS %s' % (
                n[:46], cur.get('VGPRs'), cur.get('AGPRs'), cur.get('TotalSGPRs'), cur.get('VGPRs Spill'),
                cur.get('ScratchSize [bytes/lane]'), cur.get('Occupancy [waves/SIMD]'), v))
