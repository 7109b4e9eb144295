# the hoisted x-projection of the test network's input layer (XW_XPROJ_MIN_D=1: always, 999: never) against the plain launch, in the
# sub-step cycle of bench.py: the headline shape at several d, then BASELINE configs[2] and an eighth of configs[3]
set -e
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %9.2f sub-steps/s  %.4f ms  dominant-kernel frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for dim in 20 30 40 50; do for x in 1 999; do
  echo "== d=$dim N=4096 N_t=32  XW_XPROJ_MIN_D=$x"; XW_XPROJ_MIN_D=$x timeout -k 10 120 python bench.py --dim $dim --no-cpu-baseline --train-iters 0 --no-solo --no-strong 2>/dev/null | line
done; done
for x in 1 999; do
  echo "== configs[2] d=50 N_t=64 16384  XW_XPROJ_MIN_D=$x"; XW_XPROJ_MIN_D=$x timeout -k 10 200 python bench.py --dim 50 --n_t 64 --n_r 16384 --n_b 16384 --no-cpu-baseline --train-iters 0 --no-solo --no-strong --steps 20 --repeats 3 2>/dev/null | line
  echo "== configs[3]/8 d=100 N_t=128 8192  XW_XPROJ_MIN_D=$x"; XW_XPROJ_MIN_D=$x timeout -k 10 200 python bench.py --dim 100 --n_t 128 --n_r 8192 --n_b 8192 --no-cpu-baseline --train-iters 0 --no-solo --no-strong --steps 20 --repeats 3 2>/dev/null | line
done
for x in 1 999; do
  echo "== d=20 N_t=64 16384  XW_XPROJ_MIN_D=$x"; XW_XPROJ_MIN_D=$x timeout -k 10 200 python bench.py --dim 20 --n_t 64 --n_r 16384 --n_b 16384 --no-cpu-baseline --train-iters 0 --no-solo --no-strong --steps 20 --repeats 3 2>/dev/null | line
done
echo "== defaults (from d = 45): headline, configs[2]"
timeout -k 10 120 python bench.py --no-cpu-baseline --train-iters 0 --no-solo --no-strong 2>/dev/null | line
timeout -k 10 200 python bench.py --dim 50 --n_t 64 --n_r 16384 --n_b 16384 --no-cpu-baseline --train-iters 0 --no-solo --no-strong --steps 20 --repeats 3 2>/dev/null | line
