# sweeps A + boundary at wave priority 1 (drop 2) against 0 (drop 3) at other problem sizes (bash tools/ab_cfg_prio.sh)
B="python bench.py --no-cpu-baseline --train-iters 0 --gpus 1 --no-solo"
one() { XW_PRIO_DROP_A=$1 $B "${@:3}" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$2 A=$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do for a in 2 3; do
one $a cfg3share_8192x32 --dim 100 --global-paths 8192
one $a cfg2share_2048x64 --dim 50 --n_t 64 --global-paths 2048
one $a d20_8192x32 --global-paths 8192
one $a d20_2048x32 --global-paths 2048
done; done
