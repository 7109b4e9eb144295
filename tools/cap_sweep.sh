# headline bench under different block caps of the test network / wave priorities of the stepper launches (bash tools/cap_sweep.sh)
# XW_V_BLOCKS / XW_V_BLOCKS_DISC: block caps of the test network in the generator / discriminator sub-step (of 512 slots)
# XW_PRIO_DROP_A / _G / _F / _X: wave priority 3 - drop for sweeps A + boundary / generator forward / discriminator forward / x-only sweep
# XW_EARLY_SLAB_SUM: sum the slabs of sweeps A + boundary beside sweep B (1, default) or inside the update (0)
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo --steps 90 --warmup 12"
run() { env "$@" $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
run XW_PRIO_DROP_A=2
run XW_PRIO_DROP_A=3
run XW_PRIO_DROP_A=3 XW_EARLY_SLAB_SUM=0
done
for g in 352 368 400; do run XW_V_BLOCKS=$g; done
for dsc in 400 432; do run XW_V_BLOCKS_DISC=$dsc; done
