#!/bin/bash
# rocprofv3 kernel-trace stats of the captured train step alone; $1 = output tag (default r3_step), rest = bench flags
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
T=${1:-r3_step}; shift
rm -rf $O/prof_step
rocprofv3 --kernel-trace --stats -d $O/prof_step --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline "$@" > $O/prof_step.log 2>&1
f=$(find $O/prof_step -name '*kernel_stats.csv' | head -1); cp "$f" $O/${T}_kernel_stats.csv
rm -rf $O/prof_step
tail -1 $O/prof_step.log | cut -c1-160
