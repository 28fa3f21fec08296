#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script: prof_cmd.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
T=$1; shift
rm -rf $O/prof_cmd
rocprofv3 --kernel-trace --stats -d $O/prof_cmd --output-format csv -- python3 $R/"$@" > $O/prof_cmd.log 2>&1
f=$(find $O/prof_cmd -name '*kernel_stats.csv' | head -1); cp "$f" $O/${T}_kernel_stats.csv
rm -rf $O/prof_cmd
tail -3 $O/prof_cmd.log | cut -c1-200
