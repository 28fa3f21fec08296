#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_wg5
rocprofv3 --kernel-trace --stats -d $O/prof_wg5 --output-format csv -- python3 $R/scratch/time_convs.py "conv_wgrad[enc1" > $O/prof_wg5.log 2>&1
f=$(find $O/prof_wg5 -name '*kernel_stats.csv' | head -1); grep -i "wgrad" "$f" | cut -c1-200
rm -rf $O/prof_wg5
