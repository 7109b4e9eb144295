"""HBM traffic per launch from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; unit KB) -> profiles/rNN_pmc_traffic.json

    python tools/pmc_summary.py <fetch pass: csv or output dir> <write pass: csv or output dir> <out.json>

FETCH_SIZE is doubled: gfx950 reports half the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, re, sys, collections


def short(name):
    m = re.search(r'(k_\w+)(<[^>]*>)?', name)
    return re.sub(r'\s+', '', m.group(0)) if m else None


def collect(path, counter):
    acc = collections.defaultdict(list)
    files = glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True) if os.path.isdir(path) else [path]
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k and r['Counter_Name'] == counter:
                acc[k].append(float(r['Counter_Value']))
    return acc


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 12 --warmup 3 '
                  '--no-cpu-baseline --train-iters 0 (two separate passes)',
       'workload': 'Ex4_1 cube d=20 N_r=N_b=4096 N_t=32 (bench default)',
       'note': 'FETCH_SIZE doubled (gfx950 reports half the bytes of coalesced streaming reads, MI355X_MICROARCH.md HBM '
               'section); ode kernels: average over 1-job and 2-job launches',
       'kernels': {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    fa, wa = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    out['kernels'][k] = {'FETCH_SIZE_KB_avg_per_launch': round(fa, 1), 'launches_FETCH_SIZE': len(f),
                         'WRITE_SIZE_KB_avg_per_launch': round(wa, 1), 'launches_WRITE_SIZE': len(w),
                         'hbm_bytes_per_launch_corrected': int(round((2 * fa + wa) * 1024))}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_per_launch_corrected'] for k, v in out['kernels'].items()}, indent=1))
