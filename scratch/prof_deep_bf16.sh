#!/bin/bash
# kernel durations of the bf16-native deep kernels (and the kernels they replace) from rocprofv3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd && mkdir -p /tmp/pd
WHAT=${WHAT:-down,up,wgrad} rocprofv3 --kernel-trace --stats -d /tmp/pd --output-format csv -- python3 $GRAFT_REPO_ROOT/scratch/time_deep_bf16.py > /tmp/pd/log.txt 2>&1
tail -5 /tmp/pd/log.txt
f=$(find /tmp/pd -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if any(k in n for k in ('deep_', 'wgrad_reduce', 'shadow')):
        print(f"{n[:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1000:8.1f} us")
PY
mkdir -p $GRAFT_REPO_ROOT/gpurun_out && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/deep_bf16_kernel_stats.csv
