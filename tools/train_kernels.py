"""kernels of ONE outer iteration of train()'s pipelined loop at the headline size, from a rocprofv3 --kernel-trace of tools/train_phases.py:
name, launches per outer iteration, GPU microseconds per outer iteration (busy, not elapsed):  python tools/train_kernels.py <kernel_trace.csv> [iterations]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last two train() calls of the tool run n outer iterations each: take the trailing 2 n k_disc_rec launches' span
idx = [i for i, r in enumerate(rows) if 'k_disc_rec' in r['Kernel_Name']]
lo, hi = idx[-n - 1], idx[-1]
cnt, us = collections.Counter(), collections.Counter()
for r in rows[lo + 1:hi + 1]:
    m = re.search(r'(k_\w+)', r['Kernel_Name'])
    name = m.group(1) if m else re.sub(r'\(.*', '', re.sub(r'<.*', '', r['Kernel_Name']))[:70]
    cnt[name] += 1
    us[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (int(rows[hi]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])) / 1e3 / n
print('%.1f us per outer iteration elapsed; %d launches per outer iteration, %.1f us busy' % (span, sum(cnt.values()) / n, sum(us.values()) / n))
for name, c in sorted(cnt.items(), key=lambda kv: -us[kv[0]]):
    print('%7.2f launches  %8.1f us   %s' % (c / n, us[name] / n, name))
