# Is register-file RESIDENCY what the generator sub-step loses?  The test network's forward (256 registers) and the stepper's forward
# (184) built for three waves per SIMD (168 each, by launch bounds: both spill) -- then two test-network waves AND a forward wave fit
# one SIMD -- against the shipped builds, in the headline cycle, over the test network's block cap in the generator sub-step:
#   tools/build_variant.sh lb3 xw_disc.hip "-DXW_DISC_FWD_WAVES=3"
#   tools/build_variant.sh lb3f xw_disc.hip "-DXW_DISC_FWD_WAVES=3" xw_ode.hip "-DXW_ODE_FWD_WAVES=3"
#   cp xnode_wan_pde_solver_amd/libxnwan.so _var/libxnwan_base.so; bash tools/lb3_sweep.sh
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo --no-strong --steps 90 --warmup 12"
run() { env "$@" $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'), d['kernels'].get('ode_fwd_2job'))"; }
run XW_LIBRARY=_var/libxnwan_base.so
run XW_LIBRARY=_var/libxnwan_lb3.so
for g in 384 416 448 480 512 576; do run XW_LIBRARY=_var/libxnwan_lb3f.so XW_V_BLOCKS=$g; done
for dsc in 416 448 512; do run XW_LIBRARY=_var/libxnwan_lb3f.so XW_V_BLOCKS=512 XW_V_BLOCKS_DISC=$dsc; done
run XW_LIBRARY=_var/libxnwan_base.so
