# narrow-tile thresholds (XW_NARROW_TILES=f:x:p, 16-path tiles per launch) on shards: ms per sub-step
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %9.2f sub-steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
B="python bench.py --no-cpu-baseline --train-iters 0 --no-solo --no-strong --repeats 3"
for cfg in "--dim 50 --n_t 64 --global-paths 2048" "--dim 20 --n_t 32 --global-paths 512" "--dim 20 --n_t 32 --global-paths 1024" "--dim 100 --global-paths 8192"; do
  for nt in 192:128:64 256:128:64 384:128:64 384:256:64 384:256:384; do
    echo "== $cfg   XW_NARROW_TILES=$nt"; XW_NARROW_TILES=$nt timeout -k 10 150 $B $cfg 2>/dev/null | line
  done
done
