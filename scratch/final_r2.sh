#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 300 gpurun_out/bench_full.err
for mode in eager two-graph; do
  python bench.py --force-dist --dist-mode $mode --no-roofline --no-cpu-baseline > gpurun_out/bench_dist_$mode.json 2> gpurun_out/bench_dist_$mode.err
  tail -c 300 gpurun_out/bench_dist_$mode.err
done
python bench.py --no-extra --no-roofline --no-cpu-baseline --no-graph > gpurun_out/bench_eager.json 2>&1
for f in bench_full bench_dist_eager bench_dist_two-graph bench_eager; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    l=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, l['ms_per_step'], l['value'], l['config'].get('launch'), l['config'].get('grad_buckets_bytes'), [ (e['label'], e['ms_per_step']) for e in l.get('extra',[])])
except Exception as e:
    print(f, 'ERR', e)
PY
done
