for vbd in 384 416 448 480 512; do echo "disc blocks $vbd: $(XW_V_BLOCKS_DISC=$vbd python tools/step_times.py 60 2>&1 | tail -1)"; done
for vb in 352 368 400 416; do echo "gen blocks $vb: $(XW_V_BLOCKS=$vb python tools/step_times.py 60 2>&1 | tail -1)"; done
