"""GPU busy time and the heaviest kernels of the steady part of a rocprofv3 kernel trace: python tools/cfg5_trace_summary.py trace.csv"""
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(0.4 * len(rows)):]          # the last 60 % of the trace = steady iterations
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
busy, n, ev = collections.Counter(), collections.Counter(), []
for r in rows:
    k = re.search(r'(k_\w+|\w+)', r['Kernel_Name']).group(0)[:40]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy[k] += e - s
    n[k] += 1
    ev += [(s, 1), (e, -1)]
ev.sort()
cover, depth, last = 0, 0, None
for t, d in ev:
    if depth > 0:
        cover += t - last
    depth += d
    last = t
print('window %.1f ms, %d kernels, GPU busy (union of kernel intervals) %.1f ms = %.0f %%' % ((t1 - t0) / 1e6, len(rows), cover / 1e6, 100.0 * cover / (t1 - t0)))
for k, v in busy.most_common(16):
    print('%-42s %6d launches %8.2f ms  avg %6.1f us' % (k, n[k], v / 1e6, v / n[k] / 1e3))
