#!/bin/bash
# round-3 final evidence: launch modes of the step on one box, then the profile set (scratch/profile_final.sh)
F="--no-extra --no-cpu-baseline --no-roofline --steps 200 --warmup 20"
for mode in "" "--no-graph" "--force-dist" "--force-dist --dist-graph"; do
  python bench.py $F $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MODE [$mode]', d['ms_per_step'], d['config'].get('launch'))"
done
bash scratch/profile_final.sh r3
