#!/bin/bash
# round 6, second pass (after the front-end kernel and the bf16 Linear products changed): stats of the configurations they
# touch + the front-end's PMC traffic.  usage: profile_r6b.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r6
mkdir -p $O
for spec in "07_8l_bf16:--arch speccnn8l1_bn --dim-z 512 --dtype bf16" "07_audio:--input audio" \
            "08_8l_bf16_step_only:--no-roofline --arch speccnn8l1_bn --dim-z 512 --dtype bf16" "08_audio_step_only:--no-roofline --input audio"; do
  n=${spec%%:*}; fl=${spec#*:}
  rm -rf $O/prof_tmp
  rocprofv3 --kernel-trace --stats -d $O/prof_tmp --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline $fl > $O/prof_${TAG}_$n.log 2>&1
  f=$(find $O/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" $O/${TAG}_${n}_kernel_stats.csv
  tail -1 $O/prof_${TAG}_$n.log | cut -c1-160
done
rm -rf $O/prof_tmp $O/pmc_fe_fetch $O/pmc_fe_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fe_fetch --output-format csv -- python3 $R/profiles/pmc_frontend.py run > $O/pmc_fe_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_fe_write --output-format csv -- python3 $R/profiles/pmc_frontend.py run > $O/pmc_fe_write.log 2>&1
(cd $R && python3 profiles/pmc_frontend.py traffic gpurun_out/pmc_fe_fetch gpurun_out/pmc_fe_write > gpurun_out/${TAG}_traffic_frontend.json 2> gpurun_out/${TAG}_traffic_frontend.err)
cat $O/${TAG}_traffic_frontend.json; tail -3 $O/${TAG}_traffic_frontend.err
rm -rf $O/pmc_fe_fetch $O/pmc_fe_write
