#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, 3 counters per pass, no tracing) over a small driver script; per-launch averages
# per kernel as JSON on stdout.    usage (GPU box, repo root):  bash tools/sq_counters.sh tools/ode_only.py > out.json
set -e
R=$PWD; T=${1:-tools/ode_only.py}; O=$R/gpurun_out/sq_$(basename $T .py); rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/p$i -- python3 $R/$T > $O/p$i.log 2>&1 || echo "pass $i failed" >&2
done
cd $R
python3 - "$O" "$T" <<'PY'
import csv, glob, collections, json, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+)(<[^>]*>)?', r['Kernel_Name'])
        if not m:
            continue
        name = re.sub(r'\s+', '', m.group(0))
        acc[name][r['Counter_Name']] += float(r['Counter_Value']); cnt[name][r['Counter_Name']] += 1
out = {'command': 'bash tools/sq_counters.sh ' + sys.argv[2] + '   (rocprofv3 --pmc, 3 SQ counters per pass, five passes)',
       'note': 'per launch, summed over the chip; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES in units of 4 clocks; '
               'SQ_VALU_MFMA_BUSY_CYCLES in clocks summed over the SIMDs (64 per v_mfma_f64_16x16x4, 16 reported per v_mfma_f64_4x4x4)',
       'kernels': {k: {c: round(acc[k][c] / cnt[k][c]) for c in sorted(acc[k])} for k in sorted(acc)}}
for k, v in out['kernels'].items():
    if v.get('SQ_INSTS_MFMA'):
        v['valu_per_mfma'] = round(v.get('SQ_INSTS_VALU', 0) / v['SQ_INSTS_MFMA'], 2)
    v['launches'] = max(cnt[k].values())
print(json.dumps(out, indent=1))
PY
