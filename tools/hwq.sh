# the headline bench under different numbers of hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES, default 4): bash tools/hwq.sh
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo --steps 90 --warmup 12"
for q in 4 2 3 5 6 8 4; do
GPU_MAX_HW_QUEUES=$q $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('GPU_MAX_HW_QUEUES=$q', d['ms_per_step'], d['value'])"
done
