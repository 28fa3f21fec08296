#!/bin/bash
# run-to-run spread of the eager launch modes
F="--no-extra --no-cpu-baseline --no-roofline --steps 100 --warmup 20"
for i in 1 2 3; do
for mode in "--no-graph" "--force-dist"; do
  python bench.py $F $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MODE [$mode]', d['ms_per_step'])"
done
done
