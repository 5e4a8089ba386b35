#!/usr/bin/env python3
"""tools/write_perf_matrix.py <round>: gpurun_out/matrix_<round>/ (tools/perf_matrix.sh <round>) -> profiles/<round>_perf_matrix.md.
A log that holds a Python traceback is an ERROR here (exit 1): a matrix with a crash where a measurement should be is not evidence."""
import json, os, sys
ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r04'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(ROOT, 'gpurun_out', 'matrix_' + ROUND) + '/'
out = ['# Performance matrix of %s (one MI355X; `tools/perf_matrix.sh %s`)\n' % (ROUND, ROUND),
       'Every JSON block below is the line `bench.py` printed (default flags unless stated), trimmed to the measured fields.\n']


def trim(x):
    r = x['roofline']
    o = {'value_G': round(x['value'] / 1e9, 2), 'us_per_step': round(x['ms_per_step'] * 1e3, 2), 'n_gpus': x['n_gpus'], 'steps': x['steps'],
         'resets_in_timed_region': x['resets_in_timed_region'], 'roofline_frac': r['frac'], 'roofline_bytes_model': r.get('bytes_model'),
         'launch_period_us_event_pair': round(r.get('launch_period_ms_event_pair', r.get('kernel_ms_avg', 0)) * 1e3, 3), 'frac_event_pair': r.get('frac_event_pair'),
         'traffic_bytes': r.get('traffic'), 'traffic_frac': r.get('traffic_frac', r.get('frac_of_peak_on_measured_traffic')),
         'floor_us': (r.get('floor') or {}).get('floor_us'), 'frac_of_floor': r.get('frac_of_floor'), 'value_contract_G': round(x.get('value_contract', x['value']) / 1e9, 2),
         'prepared_episodes': x.get('prepared_episodes')}
    o['kernel'] = r.get('kernel')
    if 'gather' in x:
        o['gather'] = {k: x['gather'][k] for k in ('ms', 'GBps_into_root', 'payload_bytes_per_rank', 'ranks', 'backend')}
    if 'fused_rollout' in x:
        o['fused_G'] = round(x['fused_rollout']['value'] / 1e9, 1)
        o['fused_supplied_actions_G'] = round(x['fused_rollout'].get('with_supplied_actions', {}).get('value', 0) / 1e9, 1)
    if 'staggered_resets' in x:
        o['staggered_inline_G'] = round(x['staggered_resets']['inline_resets']['value'] / 1e9, 2)
        o['staggered_prepared_G'] = round(x['staggered_resets']['prepared_next_episodes']['value'] / 1e9, 2)
    if 'api_mode' in x:
        o['api_mode_G'] = round(x['api_mode']['value'] / 1e9, 3)
    if 'api_mode_lidar' in x:
        o['api_mode_lidar_G'] = round(x['api_mode_lidar']['value'] / 1e9, 3)
    if 'lidar' in x:
        o['fused_lidar_us_per_step'] = {k: round(v['ms_per_step'] * 1e3, 2) for k, v in x['lidar'].items() if isinstance(v, dict)}
    if 'repeats' in x:
        o['repeats_us_device'] = {k: round(v * 1e3, 3) for k, v in x['repeats']['ms_per_step_device'].items()}
    if 'cold_region' in x:
        o['cold_region_us_per_step'] = round(x['cold_region']['ms_per_step'] * 1e3, 2)
    o['frac_contract_region'] = r.get('frac_contract_region', r.get('frac_wall'))
    if 'c1_single_env' in x:
        o['c1_steps_per_s'] = {k: v['value'] for k, v in x['c1_single_env'].items()}
    if 'cpu_baseline' in x:
        o['cpu_G_16_threads'] = round(x['cpu_baseline']['value'] / 1e9, 3)
        o['cpu_G_1_thread'] = round(x['cpu_baseline']['one_core_value'] / 1e9, 4)
    return o


for f, title in [('bench_C2', 'C2 `python bench.py`'), ('bench_C3', 'C3 `--workload C3`'), ('bench_C4', 'C4 `--workload C4`'), ('bench_C5', 'C5 `--workload C5`'),
                 ('bench_C2_driver_style', 'C2 driver-style `--steps 20 --warmup 5` (eager launches below 100 steps)'),
                 ('bench_X1', 'X1 FireWall hard (`--workload X1`: 3 000 untimed eager steps first, so that the handle has adapted; 1.5 % of the envs die per step)'),
                 ('bench_X2', 'X2 FenceRestriction hard'), ('bench_X3', 'X3 Crate hard'),
                 ('bench_2rank_C4', '2 ranks on ONE GPU (rehearsal of the N > 1 path; the ranks share the GPU, so per-rank rates halve): `python bench.py --gpus 2 --dist-backend gloo --single-device --workload C4 --steps 400`'),
                 ('bench_2rank_C5', '2 ranks on ONE GPU: `... --workload C5 --steps 400`')]:
    try:
        x = json.load(open(d + f + '.json'))
    except Exception:
        continue
    out.append('## %s\n\n```json\n%s\n```\n' % (title, json.dumps(trim(x))))
for f, title in [('reset_time.log', 'Reset launches (`tools/reset_time.py`, HIP events, eager; dedicated kernel where it applies)'),
                 ('reset_time_general.log', 'Reset launches, general kernel only (`NGW_FAST_RESET=0`)'),
                 ('short_run.log', 'A 20-step region fence to fence, eager vs one graph replay (`tools/short_run.py`)'),
                 ('stamps_reset.log', 'In-kernel timelines of the new-episode kernel (`tools/stamp_reset.py C3 C5 X1`, stamps build)'),
                 ('api_delta.log', 'Host API at 65 536 envs, delta refresh (`tools/api_mode_rate.py`)'),
                 ('api_full_copy.log', 'Host API at 65 536 envs, whole observation every step (`NGW_HOST_DELTA=0`)'),
                 ('churn_X1.log', 'FireWall hard, eager loop: cadence the handle adapts to, resets that miss their prepared episode (`tools/churn_probe.py X1`)'),
                 ('lidar.log', 'LidarInFront observation (`tools/lidar_rate.py`, C2, 8 beams)'),
                 ('adapter.log', 'Single-env adapter (`tools/adapter_latency.py`)'), ('api.log', 'Host API by batch size (`tools/api_latency.py`)'),
                 ('stamps.log', 'In-kernel timelines of one step launch (`tools/stamp_timeline.py`, stamps build)'),
                 ('stamps_staged.log', 'The same for the step kernel that stages the maps through LDS, at C2 (`NGW_NOSTAGE=0`)'),
                 ('ab_stage.log', '10 x 10: map staged through LDS against map read in place, by batch size (`tools/ab_stage.sh`, 1000-step graph replay)')]:
    if os.path.exists(d + f):
        text = ''.join(ln for ln in open(d + f) if 'amdgpu.ids' not in ln).strip()
        if 'Traceback (most recent call last)' in text:
            sys.exit('%s holds a traceback: fix the run, do not commit the matrix\n%s' % (f, text[-1500:]))
        out.append('## %s\n\n```\n%s\n```\n' % (title, text))
open(os.path.join(ROOT, 'profiles', ROUND + '_perf_matrix.md'), 'w').write('\n'.join(out))
print('wrote profiles/%s_perf_matrix.md' % ROUND)
