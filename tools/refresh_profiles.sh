#!/bin/bash
# Regenerate the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'
# Outputs land in gpurun_out/refresh/; copy the summaries into profiles/ afterwards (tools/pmc_summary.py for the PMC passes).
set -e -o pipefail
R=$PWD; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 60 --warmup 6 --repeats 3 --no-cpu-baseline --train-iters 0 --no-solo --no-strong"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py $ARGS > $O/rocprof_bench_line.json 2> $O/prof_default.log
echo default-mode profile done
XW_STREAMS=0 XW_GRAPHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py $ARGS > $O/rocprof_bench_line_serial.json 2> $O/prof_serial.log
echo serial profile done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --train-iters 0 --no-solo --no-strong > $O/pmc_fetch.json 2> $O/pmc_fetch.log
echo fetch pass done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --train-iters 0 --no-solo --no-strong > $O/pmc_write.json 2> $O/pmc_write.log
echo write pass done
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_summary.log
cp $O/pmc_traffic.json profiles/${XW_ROUND:-r02}_pmc_traffic.json        # bench.py reads roofline.traffic from the newest r*_pmc_traffic.json
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
echo bench done
