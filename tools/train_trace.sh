# kernel trace of train() at the headline size -> gpurun_out/tt/summary.txt   (bash tools/train_trace.sh)
set -e
R=$PWD; O=$R/gpurun_out/tt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/tools/train_trace.py 60 > $O/log.txt 2>&1
cd $R
python3 tools/train_trace_summary.py $(ls $O/p/*/*_kernel_trace.csv | head -1) 40 $XW_TRACE_ALL > $O/summary.txt
rm -rf $O/p
