# kernel trace of a few outer iterations on a ball domain (config 5 size): where the GPU time of an iteration goes
set -e
R=$PWD; O=$R/gpurun_out/c5trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/train_cfg5.py ${1:-NSphere_THourglass} 10 > $O/log.txt 2>&1
cd $R
python3 tools/cfg5_trace_summary.py $(ls $O/p/*/*_kernel_trace.csv | head -1) > $O/summary.txt
python3 tools/cfg5_group_timeline.py $(ls $O/p/*/*_kernel_trace.csv | head -1) > $O/group_timeline.txt
rm -rf $O/p
cat $O/summary.txt; tail -1 $O/log.txt
