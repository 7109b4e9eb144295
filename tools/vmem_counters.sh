#!/bin/bash
# vector-memory / TLB counters of the duo parameter sweep at 1, 2, 3 concurrent jobs (tools/sweep_jobs_only.py; launches told
# apart by grid size) -- rocprofv3 --pmc passes, per launch, summed over the chip
set -e
R=$PWD; O=$R/gpurun_out/vm; rm -rf $O; mkdir -p $O
T=${1:-tools/sweep_jobs_only.py}; KERN=${2:-k_ode_bwd_duo}
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/p$i -- python3 $R/$T > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - "$KERN" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/vm/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name']:
            acc[int(r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for gsz in sorted(acc):
    print('grid', gsz, {k: round(sum(v) / len(v)) for k, v in sorted(acc[gsz].items())})
PY
