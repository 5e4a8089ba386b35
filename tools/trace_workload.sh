#!/bin/bash
# Per-dispatch kernel durations of a workload: tools/trace_workload.sh C5 [steps]  -> gpurun_out/trace_<W>/ + a short summary
W=${1:-C5}; K=${2:-250}
OUT=gpurun_out/trace_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-stagger --workload $W --steps $K --warmup 20 > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, re, statistics as st
f = sorted(glob.glob('$OUT/**/*kernel_trace.csv', recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if 'ngw_kernel' in r['Kernel_Name']]
by = {}
for r in rows:
    m = re.search(r'ngw_kernel<\d+, (\d+),', r['Kernel_Name']).group(1)
    by.setdefault(m, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for m, d in by.items():
    ds = sorted(d)
    print('mode', m, 'n', len(d), 'median %.1f mean %.1f max %.1f us' % (st.median(d), st.mean(d), max(d)), 'top5', [round(x, 1) for x in ds[-5:]])
PY
