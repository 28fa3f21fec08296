#!/bin/bash
# SQ counters of the 1-channel end-layer kernels (one process, counters averaged per kernel name)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  rm -rf $O/pmc_c1
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_c1 --output-format csv -- python3 $R/scratch/pmc_c1.py > $O/pmc_c1.log 2>&1
  f=$(find $O/pmc_c1 -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name']
    if not any(x in n for x in ('c1_v2', 'wgrad5', 'down_q', 'up_q')): continue
    acc[n][r['Counter_Name']] += float(r['Counter_Value']); cnt[(n, r['Counter_Name'])] += 1
for n, d in acc.items():
    c = {k: v / cnt[(n, k)] for k, v in d.items()}
    print(n[:90].replace('void (anonymous namespace)::', ''))
    print('   ' + '  '.join('%s %.4g' % (k.replace('SQ_', ''), v) for k, v in sorted(c.items())))
P
done
rm -rf $O/pmc_c1
