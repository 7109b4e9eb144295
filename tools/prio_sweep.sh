# wave-priority drops of the stepper launches that are off a sub-step's critical path (engine.prio_drop), headline workload
run() { python bench.py --no-cpu-baseline --train-iters 0 --no-solo 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], o['value'], o['ms_per_step'])" "$1"; }
XW_PRIO_DROP_A=0 run "A0"
XW_PRIO_DROP_A=1 run "A1"
XW_PRIO_DROP_A=2 run "A2 (default)"
XW_PRIO_DROP_A=3 run "A3"
XW_PRIO_DROP_X=1 run "A2 X1"
XW_PRIO_DROP_F=1 run "A2 F1"
XW_PRIO_DROP_G=1 run "A2 G1"
run "A2 again"
