#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_mfma.log 2>&1
cd $R
python3 profiles/pmc_launches.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/r2_traffic.json 2> gpurun_out/r2_traffic.err
python3 profiles/pmc_launches.py mfma gpurun_out/pmc_mfma > gpurun_out/r2_mfma_util.json 2> gpurun_out/r2_mfma.err
cat gpurun_out/r2_traffic.err gpurun_out/r2_mfma.err | tail -5


rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
