#!/bin/bash
# host-side profile of the eager launch modes: cProfile over bench.py, top cumulative entries
for mode in "--no-graph" "--force-dist"; do
  echo "== $mode"
  python -m cProfile -o /tmp/prof.out bench.py --no-extra --no-cpu-baseline --no-roofline --steps 40 --warmup 10 $mode 2>/dev/null | tail -1 | cut -c1-200
  python - <<'PY'
import pstats
p = pstats.Stats('/tmp/prof.out'); p.sort_stats('tottime').print_stats(14)
PY
done
