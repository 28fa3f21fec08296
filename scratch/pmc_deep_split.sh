#!/bin/bash
# SQ counters of the split-product deep kernels (and their native fp32 counterparts): one kernel per process
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for spec in "down 2 bf16x6" "down 2 native" "up 2 bf16x6" "wgrad 2 bf16x6" "down 1 bf16x6"; do
  set -- $spec
  export WHAT=$1 LAYER=$2 MODE=$3
  rm -rf $O/pmc_ds
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_ds --output-format csv -- python3 $R/scratch/pmc_deep_split.py > $O/pmc_ds.log 2>&1
  f=$(find $O/pmc_ds -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$spec" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name']
    if not any(x in n for x in ('deep_', 'k1_')): continue
    acc[n][r['Counter_Name']] += float(r['Counter_Value']); cnt[(n, r['Counter_Name'])] += 1
for n, d in acc.items():
    c = {k: v / cnt[(n, k)] for k, v in d.items()}
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(sys.argv[2], '|', n[:70].replace('void (anonymous namespace)::', ''))
    print('   wave_cycles %.3g  parked %.2f  issue-stalled %.2f (lds %.2f)  active %.2f  mfma_busy_cycles %.3g  lds_active %.3g  bank_conflict %.3g' % (
        wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc,
        c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), c.get('SQ_LDS_IDX_ACTIVE', 0), c.get('SQ_LDS_BANK_CONFLICT', 0)))
P
done
rm -rf $O/pmc_ds
