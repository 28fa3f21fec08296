#!/bin/bash
# round-2 evidence at HEAD: kernel-trace stats of the bench configurations, step-only trace, PMC passes per launch label
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_r2_* $O/pmc_fetch $O/pmc_write $O/pmc_mfma
run() { rocprofv3 --kernel-trace --stats -d $O/prof_r2_$1 --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline "${@:2}" > $O/prof_r2_$1.log 2>&1;
        f=$(find $O/prof_r2_$1 -name '*kernel_stats.csv' | head -1); cp "$f" $O/$3_kernel_stats.csv 2>/dev/null; }
rocprofv3 --kernel-trace --stats -d $O/prof_r2_4l --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/prof_r2_4l.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_r2_8l --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --arch speccnn8l1_bn > $O/prof_r2_8l.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_r2_8l_bf16 --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --arch speccnn8l1_bn --dim-z 512 --dtype bf16 > $O/prof_r2_8l_bf16.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_r2_audio --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --input audio > $O/prof_r2_audio.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_r2_step --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline > $O/prof_r2_step.log 2>&1
for t in 4l 8l 8l_bf16 audio; do
  f=$(find $O/prof_r2_$t -name '*kernel_stats.csv' | head -1); cp "$f" $O/r2_07_${t}_kernel_stats.csv
  tail -1 $O/prof_r2_$t.log | cut -c1-200
done
f=$(find $O/prof_r2_step -name '*kernel_stats.csv' | head -1); cp "$f" $O/r2_08_4l_step_only_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_mfma.log 2>&1
cd $R
python3 profiles/pmc_launches.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/r2_traffic.json 2> gpurun_out/r2_traffic.err
python3 profiles/pmc_launches.py mfma gpurun_out/pmc_mfma > gpurun_out/r2_mfma_util.json 2> gpurun_out/r2_mfma.err
cat gpurun_out/r2_traffic.err gpurun_out/r2_mfma.err | tail -3
rm -rf $O/prof_r2_4l $O/prof_r2_8l $O/prof_r2_8l_bf16 $O/prof_r2_audio $O/prof_r2_step $O/pmc_fetch $O/pmc_write $O/pmc_mfma
