# how much of the test network's launch is the indivisible last tile?  Same kernel, same cap, time per tile unit (tiles + the 256
# gradient tiles counted twice) as the number of units per wave crosses an integer: L = 26, 29, 32, 35 at N = 4096 are 4.5, 5.0, 5.5,
# 6.0 units per wave at the generator cap (384 blocks = 1536 waves)
for L in 23 26 29 32 35 38; do AB_L=$L AB_ROUNDS=5 python tools/ab_disc.py _var/libxnwan_base.so 2>/dev/null | sed "s/^/L=$L /"; done
