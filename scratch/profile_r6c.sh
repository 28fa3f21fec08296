cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for spec in "07_4l:" "08_4l_step_only:--no-roofline"; do
  n=${spec%%:*}; fl=${spec#*:}
  rm -rf $O/prof_tmp
  rocprofv3 --kernel-trace --stats -d $O/prof_tmp --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline $fl > $O/prof_r6_$n.log 2>&1
  f=$(find $O/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" $O/r6_${n}_kernel_stats.csv
  grep -h ms_per_step $O/prof_r6_$n.log | python3 -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print('$n', d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('kernel_ms'))
    except Exception: pass"
done
rm -rf $O/prof_tmp
