#!/bin/bash
# The bench lines of a round: the driver's form (20 eager steps after 5) and the default (1000 replayed steps), both with every side measurement.
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - <<PY
import json
for f in ('bench_driver', 'bench_default'):
    d = json.load(open('$OUT/%s.json' % f))
    r = d['roofline']
    print('%s: value %.2f G  ms_per_step %.4f  frac %.3f frac_wall %.3f  kernel %s  repeats %s' % (f, d['value'] / 1e9, d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel'], d.get('repeats', {}).get('ms_per_step_device')))
    print('   fused_rollout %.1f G (%.3f us/step)  supplied %.1f G' % (d['fused_rollout']['value'] / 1e9, d['fused_rollout']['ms_per_step'] * 1e3, d['fused_rollout'].get('with_supplied_actions', {}).get('value', 0) / 1e9))
    print('   api_mode %.3f G (%.1f us)  api_mode_lidar %.3f G (%.1f us)  lidar %s' % (d['api_mode']['value'] / 1e9, d['api_mode']['ms_per_step'] * 1e3, d['api_mode_lidar']['value'] / 1e9, d['api_mode_lidar']['ms_per_step'] * 1e3,
          {k: round(v['ms_per_step'] * 1e3, 2) for k, v in d['lidar'].items() if isinstance(v, dict)}))
    print('   staggered %s  c1 %s  cold %s' % ({k: v.get('ms_per_step') for k, v in d['staggered_resets'].items() if isinstance(v, dict)}, {k: v['us_per_step'] for k, v in d['c1_single_env'].items()}, d.get('cold_region', {}).get('ms_per_step')))
PY
