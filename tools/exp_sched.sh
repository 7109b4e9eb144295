for cfg in "0 0" "0 1" "1 0" "1 1"; do
set -- $cfg
XW_FWD_FIRST_GEN=$1 XW_FWD_FIRST_DISC=$2 timeout -k 10 200 python bench.py --steps 60 --warmup 6 --train-iters 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd_first gen=$1 disc=$2:', d['value'], 'steps/s', d['ms_per_step'], 'ms')"
done
