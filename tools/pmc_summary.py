"""HBM traffic per launch from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; unit KB) -> profiles/rNN_pmc_traffic.json

    python tools/pmc_summary.py <fetch pass: csv or output dir> <write pass: csv or output dir> <out.json> [<cycle fetch pass> <cycle write pass>]

The optional second pair are the same two passes over tools/cycle_only.py (whole g,g,d cycles, nothing else): their totals
over the sub-step kernels, divided by the sub-steps they hold (3 x launches of k_disc_rec), are the `cycle` section --
HBM bytes of an average sub-step, which bench.py prints as whole_step.counter_mbytes_per_step.

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
if len(sys.argv) > 5:
    cf, cw = collect(sys.argv[4], 'FETCH_SIZE'), collect(sys.argv[5], 'WRITE_SIZE')
    # every launch of a sub-step kernel in that run belongs to a whole g,g,d cycle (the group load uses other kernels)
    sub = lambda k: re.match(r'k_(disc_fwd|disc_rec|disc_cot|disc_xproj|ode_fwd|ode_bwd|weak|bdry|gen_cot|adam|slab_sum|losses)', k) is not None  # noqa: E731
    ncyc = len(next((v for k, v in cf.items() if k.startswith('k_disc_rec')), []))
    if ncyc:
        fb = sum(sum(v) for k, v in cf.items() if sub(k)) * 1024
        wb = sum(sum(v) for k, v in cw.items() if sub(k)) * 1024
        out['cycle'] = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/cycle_only.py (two passes)', 'cycles': ncyc,
                        'substeps': 3 * ncyc, 'fetch_bytes_corrected_per_substep': int(2 * fb / (3 * ncyc)),
                        'write_bytes_per_substep': int(wb / (3 * ncyc)),
                        'hbm_bytes_per_substep_corrected': int((2 * fb + wb) / (3 * ncyc)),
                        'launches_per_cycle': {k: round(len(v) / ncyc, 2) for k, v in sorted(cf.items()) if sub(k)}}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_per_launch_corrected'] for k, v in out['kernels'].items()}, indent=1))
