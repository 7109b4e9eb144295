set -e
R=$PWD; O=$R/gpurun_out/ic; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/p$i -- python3 $R/tools/sweep_jobs_only.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/ic/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_ode_bwd_duo' in r['Kernel_Name']:
            acc[int(r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for gsz in sorted(acc):
    print('grid', gsz, 'jobs', gsz // 32768, {k: round(sum(v) / len(v)) for k, v in sorted(acc[gsz].items())})
PY
