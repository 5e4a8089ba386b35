#!/bin/bash
# End-of-round check on the GPU box (through gpurun, from the repo root; ~4 minutes): the whole -m gpu suite, smoke(), the default bench line and
# the driver's form of it.  Every step must succeed for the next to run; the two JSON lines land in gpurun_out/<round>/.
set -o pipefail
ROUND=${1:-r04}; OUT=gpurun_out/$ROUND; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite_final.log 2>&1 || { tail -20 $OUT/gpu_suite_final.log; exit 1; }
tail -2 $OUT/gpu_suite_final.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 300 python bench.py > $OUT/bench_final.log 2>&1 || { tail -5 $OUT/bench_final.log; exit 1; }
grep "^{" $OUT/bench_final.log | tail -1 > $OUT/bench_final.json
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_final.log 2>&1 || { tail -5 $OUT/bench_driver_final.log; exit 1; }
grep "^{" $OUT/bench_driver_final.log | tail -1 > $OUT/bench_driver_final.json
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for f in ('bench_final', 'bench_driver_final'):
    d = json.load(open('%s/%s.json' % (out, f))); r = d['roofline']
    print('%s: value %.2f G, %.3f us per step, frac %.3f, frac_wall %.3f, kernel %s' % (f, d['value'] / 1e9, d['ms_per_step'] * 1e3, r['frac'], r['frac_wall'], r['kernel']))
    fr = d.get('fused_rollout') or {}
    print('   repeats (device ms per step)', d['repeats']['ms_per_step_device'])
    print('   api_mode %.2f G; lidar int16 %.2f us; api_mode_lidar %.2f G; fused rollout %.1f G (one launch, inline resets: %.1f G)' % (
        d['api_mode']['value'] / 1e9, d['lidar']['int16']['ms_per_step'] * 1e3, d['api_mode_lidar']['value'] / 1e9,
        fr.get('value', 0) / 1e9, (fr.get('inline_resets_one_launch') or {}).get('value', 0) / 1e9))
PY
