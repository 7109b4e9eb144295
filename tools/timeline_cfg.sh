# kernel timeline of one g,g,d cycle for any bench.py workload:  bash tools/timeline_cfg.sh NAME --dim 50 --n_t 64 --global-paths 2048
set -e
name=$1; shift
R=$PWD; O=$R/gpurun_out/tl_$name; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/bench.py --steps 30 --warmup 6 --repeats 2 --no-cpu-baseline --train-iters 0 --no-solo --no-strong "$@" > $O/line.json 2> $O/log.txt
cd $R
python3 tools/timeline.py $(ls $O/p/*/*_kernel_trace.csv | head -1) 12 > gpurun_out/timeline_$name.txt
cat gpurun_out/timeline_$name.txt
