# round-5 profile refresh (run through gpurun from the repo root): bench under rocprofv3 (as launched + serial), PMC passes,
# final bench line, kernel times, timeline, SQ counters of the test network
set -e -o pipefail
export XW_ROUND=r05
bash tools/refresh_profiles.sh
python3 tools/kernel_times.py > gpurun_out/refresh/kernel_times.txt 2>&1
echo kernel times done
bash tools/timeline_run.sh > gpurun_out/refresh/timeline.txt 2>&1 || true
echo timeline done
bash tools/sq_counters.sh tools/disc_only.py > gpurun_out/r05_sq_disc.json || true
echo sq disc done
