# round-4 profile refresh, part B: SQ counters (stepper, both tile layouts; test network), the other BASELINE configs at full
# batch and as shares, HBM counters of configs[3], the shard sweep with narrow tiles off / on
set -e -o pipefail
R=$PWD
[ -s gpurun_out/r04_sq_ode.json ] || bash tools/sq_counters.sh tools/ode_only.py > gpurun_out/r04_sq_ode.json
echo sq ode done
[ -s gpurun_out/r04_sq_disc.json ] || bash tools/sq_counters.sh tools/disc_only.py > gpurun_out/r04_sq_disc.json
echo sq disc done
mkdir -p gpurun_out/r04_lines; bash tools/other_configs.sh > gpurun_out/r04_other_configs_summary.txt 2>&1
echo other configs done
O=$R/gpurun_out/cfg3_pmc; rm -rf $O; mkdir -p $O
ARGS="--dim 100 --global-paths 65536 --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --train-iters 0 --no-solo"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.json 2> $O/fetch.log)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.json 2> $O/write.log)
python3 tools/pmc_summary.py $O/fetch $O/write $O/cfg3_pmc_traffic.json > $O/summary.log
rm -rf $O/fetch $O/write
echo cfg3 pmc done
mkdir -p gpurun_out/r04_shards
for n in 4096 2048 1024 512; do for v in 0 1; do
  XW_NARROW=$v python3 bench.py --global-paths $n --no-cpu-baseline --train-iters 0 --no-solo > gpurun_out/r04_shards/d20_gp${n}_narrow$v.json 2> gpurun_out/r04_shards/err_${n}_$v.txt
done; echo shard $n done; done
