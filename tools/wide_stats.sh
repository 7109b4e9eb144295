# rocprofv3 kernel statistics of g,g,d cycles at the widest containers ((64,16) stepper, 128-wide test network), headline sample
#   bash tools/wide_stats.sh   -> gpurun_out/wide/stats.csv (copied to profiles/r06_rocprofv3_kernel_stats_wide.csv)
set -e
R=$PWD; O=$R/gpurun_out/wide; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export XW_CYCLE_WIDTHS=64,16,128
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/cycle_only.py 10 > $O/log.txt 2>&1
cd $R
cp $(ls $O/p/*/*_kernel_stats.csv | head -1) $O/stats.csv
rm -rf $O/p
head -14 $O/stats.csv
