set -e
R=$PWD; O=$R/gpurun_out/sq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/p$i -- python3 $R/tools/${XW_SQ_TARGET:-bwd_only.py} > $O/p$i.log 2>&1 || echo pass $i failed
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/sq/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = 'k_disc_rec' if 'k_disc_rec' in k else 'k_disc_bwd' if 'k_disc_bwd' in k else ('k_disc_fwd<true>' if '<50, true>' in k else 'k_disc_fwd<false>') if 'k_disc_fwd' in k else None
        if name:
            acc[name][r['Counter_Name']] += float(r['Counter_Value']); cnt[name][r['Counter_Name']] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print('   %-28s %14.0f per launch (%d launches)' % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
PY
