#!/bin/bash
# same-box A/B of the bf16 8-layer z=512 configuration: scratch/base against the working tree
F="--no-extra --no-cpu-baseline --no-roofline --arch speccnn8l1_bn --dim-z 512 --dtype bf16 --steps 60 --warmup 10"
for i in 1 2; do
  (cd scratch/base && python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'])")
  python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'])"
done
