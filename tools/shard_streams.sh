# the compact generator schedule (Engine._gen_front_compact: all three sweep jobs in one launch on the main stream) against the wide one
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo"
for n in 4096 3072 2560 2048 1536; do for ct in 0 100000; do
XW_COMPACT_TILES=$ct $B --global-paths $n 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('paths $n XW_COMPACT_TILES=$ct', d['value'], d['ms_per_step'])"
done; done
