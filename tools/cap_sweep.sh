# headline bench under different block caps of the test network / wave priorities of the stepper launches (bash tools/cap_sweep.sh)
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo --steps 90 --warmup 12"
run() { env "$@" $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
run XW_PRIO_DROP_A=3
run XW_PRIO_DROP_A=3 XW_PRIO_DROP_G=1
run XW_PRIO_DROP_A=3 XW_PRIO_DROP_G=2
run XW_PRIO_DROP_A=3 XW_PRIO_DROP_G=3
run XW_PRIO_DROP_A=3 XW_PRIO_DROP_G=1 XW_PRIO_DROP_F=1 XW_PRIO_DROP_X=1
run XW_PRIO_DROP_A=3 XW_PRIO_DROP_G=1 XW_V_BLOCKS_DISC=432
done
