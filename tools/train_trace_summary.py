"""what the GPU does during one outer iteration of train(), from a rocprofv3 kernel trace of tools/train_trace.py:
   python tools/train_trace_summary.py <kernel_trace.csv> [iterations to average, from the end]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
marks = [i for i, r in enumerate(rows) if 'k_disc_rec' in r['Kernel_Name'] or 'k_disc_bwd' in r['Kernel_Name']]
i0, i1 = marks[-n - 1], marks[-1]
win = rows[i0 + 1:i1 + 1]
t0, t1 = int(rows[i0]['End_Timestamp']), int(rows[i1]['End_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('window: %d outer iterations, %.3f ms each; GPU busy (union of kernel intervals) %.3f ms each = %.0f %%; %d kernels each' % (
    n, (t1 - t0) / n / 1e6, busy / n / 1e6, 100.0 * busy / (t1 - t0), len(win) // n))
tot = collections.Counter(); cnt = collections.Counter()
for r in win:
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[:70]
    tot[name] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[name] += 1
print('%-72s %8s %8s' % ('kernel', 'per it', 'us / it'))
for name, t in tot.most_common(22):
    print('%-72s %8.1f %8.1f' % (name, cnt[name] / n, t / n / 1e3))
print('%-72s %8s %8.1f' % ('(sum of kernel durations, overlapping ones counted twice)', '', sum(tot.values()) / n / 1e3))
# one iteration as a timeline (the last complete one)
j0, j1 = marks[-2], marks[-1]
tt = int(rows[j0]['End_Timestamp'])
print('--- last iteration: start end dur (us), queue, kernel')
last_e = 0
for r in rows[j0 + 1:j1 + 1]:
    s, e = (int(r['Start_Timestamp']) - tt) / 1e3, (int(r['End_Timestamp']) - tt) / 1e3
    if e - s > 6 or s - last_e > 15 or 'all' in sys.argv:
        print('%8.1f %8.1f %7.1f  q%-2s gap %6.1f  %s' % (s, e, e - s, r['Queue_Id'], s - last_e, re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[:60] if 'all' not in sys.argv else r['Kernel_Name'].replace('at::native::', '').replace('(anonymous namespace)::', '')[:150]))
    last_e = max(last_e, e)
