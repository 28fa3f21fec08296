#!/bin/bash
# SQ counters per launch label of the 4-layer fp32 step -> gpurun_out/r4_sq_counters.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export PMC_LABELS=pmc_labels_sq.json
rm -rf $O/pmc_sq_a $O/pmc_sq_b $O/pmc_sq_c
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc_sq_a --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $O/pmc_sq_b --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES -d $O/pmc_sq_c --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_sq_c.log 2>&1
(cd $R && python3 profiles/pmc_launches.py counters gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c > gpurun_out/r4_sq_counters.json 2> gpurun_out/r4_sq.err)
tail -3 $O/r4_sq.err; rm -rf $O/pmc_sq_a $O/pmc_sq_b $O/pmc_sq_c; ls -la $O/r4_sq_counters.json
