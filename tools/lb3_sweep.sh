# the test network's forward built for three waves per SIMD (168 registers, -DXW_DISC_FWD_WAVES=3) against the shipped 256-register
# build, in the headline cycle, over the block caps of the two sub-steps (of 256 CUs x 2 -- or x 3 -- block slots):
#   tools/build_variant.sh lb3 xw_disc.hip "-DXW_DISC_FWD_WAVES=3"; cp xnode_wan_pde_solver_amd/libxnwan.so _var/libxnwan_base.so; bash tools/lb3_sweep.sh
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo --steps 90 --warmup 12"
run() { env "$@" $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))"; }
for rep in 1 2; do
run XW_LIBRARY=_var/libxnwan_base.so
run XW_LIBRARY=_var/libxnwan_lb3.so
done
for g in 384 448 512 576 640 768; do run XW_LIBRARY=_var/libxnwan_lb3.so XW_V_BLOCKS=$g; done
for dsc in 416 512 640 768; do run XW_LIBRARY=_var/libxnwan_lb3.so XW_V_BLOCKS_DISC=$dsc; done
run XW_LIBRARY=_var/libxnwan_base.so
