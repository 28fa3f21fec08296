#!/bin/bash
# kernels of one replayed step under rocprofv3 (stats of the step-only bench): usage small_kernels.sh <max_us> [bench flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; MAXUS=$1; shift
rm -rf $R/gpurun_out/prof_tmp2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tmp2 --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_tmp2 -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$MAXUS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    per = int(r["Calls"]) / 37
    us = float(r["AverageNs"]) / 1e3
    if per >= 0.9 and "copyBuffer" not in r["Name"]:
        tot += per * us
        if us < float(sys.argv[2]):
            print("%5.1f x %7.1f  %s" % (per, us, r["Name"].replace("(anonymous namespace)::", "")[:90]))
print("step total (sum of kernels) %.0f us" % tot)
PY
rm -rf $R/gpurun_out/prof_tmp2
