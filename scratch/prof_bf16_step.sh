#!/bin/bash
# kernel statistics of the 8-layer z=512 bf16 step only (no roofline probes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_tmp
rocprofv3 --kernel-trace --stats -d $O/prof_tmp --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline --arch speccnn8l1_bn --dim-z 512 --dtype bf16 > $O/prof_bf16_step.log 2>&1
f=$(find $O/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" $O/${1:-r4_09}_8l_bf16_step_only_kernel_stats.csv
tail -1 $O/prof_bf16_step.log | cut -c1-200
rm -rf $O/prof_tmp
