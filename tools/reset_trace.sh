#!/bin/bash
# Reset launch durations by mask pattern: tools/reset_trace.sh
OUT=gpurun_out/reset_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/reset_rate.py > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, re, statistics as st
f = sorted(glob.glob('gpurun_out/reset_trace/**/*kernel_trace.csv', recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if re.search(r'ngw_kernel<\d+, 1,', r['Kernel_Name'])]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
for i, name in enumerate(['all envs', 'none (staging only)', 'one lane per wave', '8 lanes per wave', 'all, sync between', 'all through a mask']):
    x = d[20 * i:20 * i + 20]
    print('reset %-22s median %.1f us  (min %.1f, max %.1f)' % (name, st.median(x), min(x), max(x)))
PY
