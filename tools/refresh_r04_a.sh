# round-4 profile refresh, part A (run through gpurun from the repo root): bench under rocprofv3 (default + serial), PMC passes,
# final bench line, kernel times, timeline
set -e -o pipefail
export XW_ROUND=r04
bash tools/refresh_profiles.sh
python3 tools/kernel_times.py > gpurun_out/refresh/kernel_times.txt 2>&1
echo kernel times done
bash tools/timeline_run.sh > gpurun_out/refresh/timeline.txt 2>&1 || true
echo timeline done
