#!/bin/bash
# same-box A/B: bench.py of scratch/base (a built copy of an earlier commit: git archive <rev> | tar -x -C scratch/base, then
# python __graft_entry__.py build there) against the working tree, alternating.  usage: ab.sh [rounds] [bench flags...]
R=${1:-2}; shift
for i in $(seq $R); do
  (cd scratch/base && python bench.py --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'])")
  python bench.py --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'])"
done
