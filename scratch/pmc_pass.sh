export TMPDIR=/tmp
for spec in "wgrad enc4" "wgrad enc2" "wgrad enc1" "down enc2" "up enc4"; do
  set -- $spec
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc2_fetch_$1_$2 --output-format csv -- python3 profiles/pmc_driver.py $1 $2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc2_write_$1_$2 --output-format csv -- python3 profiles/pmc_driver.py $1 $2 > /dev/null 2>&1
done
python3 profiles/pmc_driver.py --summarise gpurun_out/pmc2_fetch_* gpurun_out/pmc2_write_*
