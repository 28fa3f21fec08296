#!/bin/bash
# round-6 evidence at HEAD: kernel-trace stats of the bench configurations, step-only traces, PMC passes per launch label.
# The default fp32 line runs the six-bf16-instruction products; the native-instruction runs are kept next to it.
# usage: profile_r6.sh [tag] [part: all|stats|pmc]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r6}; PART=${2:-all}
mkdir -p $O
if [ "$PART" = all ] || [ "$PART" = stats ]; then
  for spec in "07_4l:" "07_8l:--arch speccnn8l1_bn" "07_8l_bf16:--arch speccnn8l1_bn --dim-z 512 --dtype bf16" "07_audio:--input audio" \
              "08_4l_step_only:--no-roofline" "08_8l_step_only:--no-roofline --arch speccnn8l1_bn" \
              "08_8l_bf16_step_only:--no-roofline --arch speccnn8l1_bn --dim-z 512 --dtype bf16" \
              "09_4l_native_step_only:--no-roofline --fp32-products native" \
              "09_8l_native_step_only:--no-roofline --arch speccnn8l1_bn --fp32-products native"; do
    n=${spec%%:*}; fl=${spec#*:}
    rm -rf $O/prof_tmp
    rocprofv3 --kernel-trace --stats -d $O/prof_tmp --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline $fl > $O/prof_${TAG}_$n.log 2>&1
    f=$(find $O/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" $O/${TAG}_${n}_kernel_stats.csv
    tail -1 $O/prof_${TAG}_$n.log | cut -c1-160
  done
  rm -rf $O/prof_tmp
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
  for cfg in "4l:speccnn4l1_bn:64:fp32:bf16x6" "8l:speccnn8l1_bn:64:fp32:bf16x6" "native:speccnn4l1_bn:64:fp32:native" \
             "8l_native:speccnn8l1_bn:64:fp32:native" "8l_bf16:speccnn8l1_bn:512:bf16:native"; do
    IFS=: read name arch dz dt prod <<< "$cfg"
    export PMC_ARCH=$arch PMC_DZ=$dz PMC_DTYPE=$dt PMC_FP32_PRODUCTS=$prod PMC_LABELS=pmc_labels_$name.json
    rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_fetch_$name.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_write_$name.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- python3 $R/profiles/pmc_launches.py run > $O/pmc_mfma_$name.log 2>&1
    sfx=$([ $name = 4l ] && echo "" || echo "_$name")
    (cd $R && python3 profiles/pmc_launches.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${TAG}_traffic$sfx.json 2> gpurun_out/${TAG}_traffic$sfx.err)
    (cd $R && python3 profiles/pmc_launches.py mfma gpurun_out/pmc_mfma > gpurun_out/${TAG}_mfma_util$sfx.json 2> gpurun_out/${TAG}_mfma$sfx.err)
    tail -2 $O/${TAG}_traffic$sfx.err $O/${TAG}_mfma$sfx.err 2>/dev/null | tail -4
    rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
  done
fi
ls -la $O/${TAG}_* | head -40
