set -e
R=$PWD; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/bench.py --steps 30 --warmup 6 --no-cpu-baseline --train-iters 0 --no-solo --no-strong > $O/line.json 2> $O/log.txt
cd $R
python3 tools/timeline.py $(ls $O/p/*/*_kernel_trace.csv | head -1) 20
