# strong-scaling shards of the headline problem on one GPU, 16-path tiles only against the default narrow-tile policy (bash tools/shard_sweep.sh)
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo"
for n in 4096 2048 1024 512; do for nar in 0 1; do
XW_NARROW=$nar $B --global-paths $n 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('paths $n XW_NARROW=$nar', d['value'], d['ms_per_step'])"
done; done
