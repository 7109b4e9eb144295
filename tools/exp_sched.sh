for cfg in "1 1 1 1" "1 1 0 1" "1 1 1 0" "1 1 0 0" "0 1 1 1" "0 1 0 0" "0 0 0 0"; do
set -- $cfg
XW_GRAPHS=$1 XW_STREAMS=$2 XW_PAR_GRADX=$3 XW_SIDE_CONTRACT=$4 timeout -k 10 200 python bench.py --steps 60 --warmup 6 --train-iters 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graphs=$1 streams=$2 par_gradx=$3 side_contract=$4:', d['value'], 'steps/s', d['ms_per_step'], 'ms')"
done
