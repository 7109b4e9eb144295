set -e -o pipefail
export XW_ROUND=r03
bash tools/refresh_profiles.sh
mkdir -p gpurun_out/r03_lines
python3 bench.py --dim 50 --n_t 64 --global-paths 2048 --no-cpu-baseline --train-iters 0 > gpurun_out/r03_lines/cfg2_share.json 2> gpurun_out/r03_lines/cfg2.err
echo cfg2 done
python3 bench.py --dim 100 --global-paths 8192 --no-cpu-baseline --train-iters 0 > gpurun_out/r03_lines/cfg3_share.json 2> gpurun_out/r03_lines/cfg3.err
echo cfg3 done
for n in 4096 2048 1024 512; do python3 bench.py --global-paths $n --no-cpu-baseline --train-iters 0 --no-solo > gpurun_out/r03_lines/d20_gp$n.json 2> gpurun_out/r03_lines/gp$n.err; echo gp$n done; done
python3 tools/kernel_times.py > gpurun_out/refresh/kernel_times.txt 2>&1
echo kernel times done
