# configs[2] and configs[3] of BASELINE.json at their STATED global batch on ONE GPU, and their 1/8 shares (profiles/r04_lines/)
set -e
O=gpurun_out/${XW_ROUND:-r04}_lines; mkdir -p $O
B="python bench.py --no-cpu-baseline --train-iters 0 --gpus 1"
$B --dim 50 --n_t 64 --global-paths 16384  > $O/cfg2_d50_16384x64_1gpu.json   2> $O/cfg2.log
$B --dim 100 --global-paths 65536          > $O/cfg3_d100_65536x32_1gpu.json  2> $O/cfg3.log
$B --dim 50 --n_t 64 --global-paths 2048   > $O/cfg2_share_2048x64_1gpu.json   2>> $O/cfg2.log
$B --dim 100 --global-paths 8192           > $O/cfg3_share_8192x32_1gpu.json   2>> $O/cfg3.log
for f in $O/*.json; do python - $f <<'PY'
import json, sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], o['value'], 'steps/s', o['ms_per_step'], 'ms/step', 'whole-step fp64', o['whole_step']['frac_fp64_matrix_peak'],
      'dominant', o['roofline']['kernel'], o['roofline']['frac'], 'solo', o['roofline'].get('solo_full_grid', {}).get('frac'))
PY
done
