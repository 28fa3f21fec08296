#!/bin/bash
# kernel + memory-copy sequence of ONE replayed step (rocprofv3 kernel trace, last graph replay): gpurun_out/<tag>_seq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
T=${1:-r3_seq}; shift
rm -rf $O/prof_seq
rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_seq --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-roofline --steps 3 --warmup 2 "$@" > $O/prof_seq.log 2>&1
python3 - "$O" "$T" <<'PY'
import csv, glob, sys
O, T = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(O + '/prof_seq/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for f in glob.glob(O + '/prof_seq/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'MEMCPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', '')))
rows.sort()
# the last step: from the last zero_grad fill back to the end
idx = [i for i, r in enumerate(rows) if 'adam4' in r[2]]
lo = idx[-2] + 1 if len(idx) >= 2 else 0
hi = idx[-1] + 1
with open(f'{O}/{T}_seq.txt', 'w') as out:
    prev_end = None
    for s, e, n in rows[lo:hi]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        out.write(f'{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {n[:150]}\n')
        prev_end = e
    out.write(f'span {(rows[hi - 1][1] - rows[lo][0]) / 1e3:.1f} us, {hi - lo} launches\n')
PY
rm -rf $O/prof_seq
tail -2 $O/${T}_seq.txt
