#!/bin/bash
# round-3 evidence at HEAD: kernel-trace stats of the bench configurations, step-only trace, PMC passes per launch label
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r3}
rm -rf $O/prof_${TAG}_* $O/pmc_fetch $O/pmc_write $O/pmc_mfma
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_4l --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/prof_${TAG}_4l.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_8l --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --arch speccnn8l1_bn > $O/prof_${TAG}_8l.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_8l_bf16 --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --arch speccnn8l1_bn --dim-z 512 --dtype bf16 > $O/prof_${TAG}_8l_bf16.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_audio --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --input audio > $O/prof_${TAG}_audio.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_step --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline > $O/prof_${TAG}_step.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_step8 --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline --arch speccnn8l1_bn > $O/prof_${TAG}_step8.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_stepb --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline --arch speccnn8l1_bn --dim-z 512 --dtype bf16 > $O/prof_${TAG}_stepb.log 2>&1
for t in 4l 8l 8l_bf16 audio; do
  f=$(find $O/prof_${TAG}_$t -name '*kernel_stats.csv' | head -1); cp "$f" $O/${TAG}_07_${t}_kernel_stats.csv
  tail -1 $O/prof_${TAG}_$t.log | cut -c1-200
done
f=$(find $O/prof_${TAG}_step -name '*kernel_stats.csv' | head -1); cp "$f" $O/${TAG}_08_4l_step_only_kernel_stats.csv
f=$(find $O/prof_${TAG}_step8 -name '*kernel_stats.csv' | head -1); cp "$f" $O/${TAG}_08_8l_step_only_kernel_stats.csv
f=$(find $O/prof_${TAG}_stepb -name '*kernel_stats.csv' | head -1); cp "$f" $O/${TAG}_08_8l_bf16_step_only_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_mfma.log 2>&1
cd $R
python3 profiles/pmc_launches.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${TAG}_traffic.json 2> gpurun_out/${TAG}_traffic.err
python3 profiles/pmc_launches.py mfma gpurun_out/pmc_mfma > gpurun_out/${TAG}_mfma_util.json 2> gpurun_out/${TAG}_mfma.err
cat gpurun_out/${TAG}_traffic.err gpurun_out/${TAG}_mfma.err | tail -3
rm -rf $O/prof_${TAG}_4l $O/prof_${TAG}_8l $O/prof_${TAG}_8l_bf16 $O/prof_${TAG}_audio $O/prof_${TAG}_step $O/prof_${TAG}_step8 $O/prof_${TAG}_stepb $O/pmc_fetch $O/pmc_write $O/pmc_mfma
