#!/bin/bash
# Kernel-time split of a workload (default X1) in bench form: tools/x1_stats.sh [workload]
W=${1:-X1}; OUT=gpurun_out/${ROUND:-r04}/stats_$W; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-side --steps 1000 --warmup 100 --workload $W > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
grep '^{' $OUT/run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9,'G', d['ms_per_step']*1e3,'us/step', d.get('prepared_episodes'))"
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:8]:
    print('%-70s calls %6s avg %9.2f us total %9.2f ms  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
find $OUT -name "*.db" -delete 2>/dev/null
