# train() at the headline configuration with the reference's acceptance rule as a `stop` hook (what its main.py passes):
python tools/train_phases.py 100 stop 2>&1 | grep -v "^ *[0-9]* .*{built-in\|^$" | head -12
