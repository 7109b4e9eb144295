# round-6 profile refresh (run through gpurun from the repo root): bench under rocprofv3 (as launched + serial), PMC passes over
# bench.py and over whole cycles only (tools/cycle_only.py -> the `cycle` section), final bench line, kernel times, timeline, SQ
# counters of the test network, train() kernel trace
set -e -o pipefail
export XW_ROUND=r06
R=$PWD; O=$R/gpurun_out/refresh
bash tools/refresh_profiles.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_cycle_fetch -- python3 $R/tools/cycle_only.py 10 > $O/pmc_cycle_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_cycle_write -- python3 $R/tools/cycle_only.py 10 > $O/pmc_cycle_write.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_cycle_fetch $O/pmc_cycle_write > $O/pmc_summary.log
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
echo cycle passes done
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
echo bench with the cycle section done
python3 tools/kernel_times.py > $O/kernel_times.txt 2>&1
echo kernel times done
bash tools/timeline_run.sh > $O/timeline.txt 2>&1 || true
echo timeline done
bash tools/sq_counters.sh tools/disc_only.py > gpurun_out/r06_sq_disc.json || true
echo sq disc done
XW_TRACE_ALL=all bash tools/train_trace.sh || true
echo train trace done
