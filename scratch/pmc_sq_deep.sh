#!/bin/bash
# usage: scratch/pmc_sq_deep.sh <tag> [fp32|bf16]   -> gpurun_out/pmcsqd_<tag>.txt  (SQ counters of the deep-layer kernels)
export TMPDIR=/tmp
tag=$1; dt=${2:-fp32}
rm -rf gpurun_out/pmcsqd_${tag}_*
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmcsqd_${tag}_a --output-format csv -- python3 profiles/pmc_deep.py $dt > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d gpurun_out/pmcsqd_${tag}_b --output-format csv -- python3 profiles/pmc_deep.py $dt > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_LDS_UNALIGNED_STALL -d gpurun_out/pmcsqd_${tag}_c --output-format csv -- python3 profiles/pmc_deep.py $dt > /dev/null 2>&1
python3 - <<PY > gpurun_out/pmcsqd_${tag}.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcsqd_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'deep_' not in n and 'k1_' not in n:
            continue
        acc[n[:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(acc.items()):
    print(k)
    for n, v in sorted(c.items()):
        print(f'   {n:32s} {sum(v)/len(v):16.1f}  (n={len(v)})')
PY
cat gpurun_out/pmcsqd_${tag}.txt
