#!/bin/bash
# copy what tools/refresh_profiles.sh (+ kernel_times.py, sq_counters.sh) left under gpurun_out/ into profiles/ (run here, after
# the gpurun call has merged its outputs):   bash tools/collect_profiles.sh [r02]
set -e
R=${1:-r02}; O=gpurun_out/refresh
cp "$(ls -t $O/prof_default/runc/*_kernel_stats.csv | head -1)" profiles/${R}_rocprofv3_kernel_stats.csv
cp "$(ls -t $O/prof_serial/runc/*_kernel_stats.csv | head -1)" profiles/${R}_rocprofv3_kernel_stats_serial.csv
cp $O/rocprof_bench_line.json profiles/${R}_rocprofv3_bench_line.json
cp $O/rocprof_bench_line_serial.json profiles/${R}_rocprofv3_bench_line_serial.json
cp $O/bench_final.json profiles/${R}_bench_final_1gpu.json
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
[ -f $O/kernel_times.txt ] && grep -v amdgpu.ids $O/kernel_times.txt > profiles/${R}_kernel_times.txt
[ -f gpurun_out/${R}_sq_ode.json ] && cp gpurun_out/${R}_sq_ode.json profiles/${R}_sq_counters_ode.json
[ -f gpurun_out/${R}_sq_disc.json ] && cp gpurun_out/${R}_sq_disc.json profiles/${R}_sq_counters_disc.json
(head -3 profiles/${R}_timeline_cycle.txt; python3 tools/timeline.py "$(ls -t $O/prof_default/runc/*_kernel_trace.csv | head -1)" 30) > /tmp/tl.txt && cp /tmp/tl.txt profiles/${R}_timeline_cycle.txt
python3 - "$O/bench_final.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench: %.1f steps/s, %.4f ms/step, roofline %.4f as launched / %.4f solo, whole step %.4f, reuse opt-in %.1f, train rel-L2 %.5f after %d' % (
    j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['solo_full_grid']['frac'], j['whole_step']['frac_fp64_matrix_peak'],
    j['extras']['steps_per_s_with_test_net_reuse_optin'], j['extras']['train']['rel_l2_heldout_16384'], j['extras']['train']['outer_iterations']))
PY
