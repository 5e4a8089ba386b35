#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into the committed summaries under profiles/:
   profiles/<tag>_kernel_stats.md   rocprofv3 --kernel-trace --stats summaries (step mode and fused rollout)
   profiles/<tag>_pmc.md            HBM traffic counters incl. the FETCH_SIZE calibration on a known byte count
   profiles/pmc_traffic.json        per-launch HBM bytes that bench.py reports as roofline.traffic
Counters are KiB; gfx950 correction per MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reads wide coalesced reads by 2x -
the factor is re-measured here on the staging-only kernel whose byte count is known exactly."""
import csv
import glob
import json
import os
import statistics as st
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def kernel_mode(name):
    """MODE template argument of ngw_kernel<MAPMODE, MODE, LIDAR, EXT>: 0 step, 1 reset, 2 rollout, 3 refill"""
    import re
    m = re.search(r'ngw_kernel<\d+, (\d+),', name)
    return int(m.group(1)) if m else -1


def kernel_rows(path, modes):
    return [r for r in csv.DictReader(open(path)) if kernel_mode(r['Kernel_Name']) in modes]


out = ['# rocprofv3 --kernel-trace --stats summaries (%s)\n' % tag,
       'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-stagger --steps 400 --warmup 100 [--mode rollout]`',
       '(workload C2: NovelGridworld-Pogostick-v1, 65 536 envs, 10x10, autoreset H=100).\n']
for mode in ('step', 'rollout'):
    f = one('stats_%s/**/*kernel_stats.csv' % mode)
    if not f:
        continue
    out.append('## %s mode - kernel_stats.csv\n' % mode)
    out.append('| kernel | calls | total ns | average ns | % | min ns | max ns |')
    out.append('|---|---|---|---|---|---|---|')
    for r in csv.DictReader(open(f)):
        name = r['Name'] if len(r['Name']) < 90 else r['Name'][:87] + '...'
        out.append('| `%s` | %s | %s | %.1f | %s | %s | %s |' % (name, r['Calls'], r['TotalDurationNs'], float(r['AverageNs']),
                                                              r['Percentage'], r['MinNs'], r['MaxNs']))
    rows = kernel_rows(one('stats_%s/**/*kernel_trace.csv' % mode), (0, 1) if mode == 'step' else (1, 2))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    out.append('')
    if mode == 'step':
        steps = d[1:]                                  # first launch is the reset-all
        cut = 3 * st.median(steps)                                   # every env hits the horizon H = 100 together: 1 launch in 100
        resets = [x for x in steps if x > cut]
        normal = [x for x in steps if x <= cut]
        big = sorted(normal)[-3:]
        out.append('Per-dispatch (kernel_trace.csv): reset-all launch %.1f us; %d step launches: median %.2f us, mean %.2f us; '
                   '%d of them hit the horizon (every env resets in that launch): %s us; the other %d: mean %.2f us, '
                   'three slowest %s us. VGPR %s, SGPR %s, workgroup %s, grid %s.'
                   % (d[0], len(steps), st.median(steps), st.mean(steps), len(resets), ', '.join('%.1f' % x for x in resets),
                      len(normal), st.mean(normal), ', '.join('%.1f' % x for x in big), rows[1]['VGPR_Count'],
                      rows[1]['SGPR_Count'], rows[1]['Workgroup_Size_X'], rows[1]['Grid_Size_X']))
        timed = steps[100:500]
        out.append('')
        out.append('The 400 TIMED launches only (after the 100 warm-up launches; this is what `bench.py` brackets with its HIP event pair): '
                   'mean %.2f us, median %.2f us, max %.1f us.' % (st.mean(timed), st.median(timed), max(timed)))
    else:
        out.append('Per-dispatch: reset-all %.1f us; warm-up rollout (100 steps) %.1f us; timed rollout (400 steps) %.1f us = %.3f us per batched step.'
                   % (d[0], d[1], d[2], d[2] / 400))
    out.append('')
for run, title in (('stagger_inline', 'inline resets'), ('stagger_prefetch', 'prepared next episodes, refill every 32 steps (NGW_PREFETCH=32)')):
    f = one('stats_%s/**/*kernel_stats.csv' % run)
    if not f:
        continue
    out.append('## staggered episode ends - %s\n' % title)
    out.append('Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/stagger_rate.py` (C2; first a synchronized '
               'run, then step_count offsets e * 7919 % 100: ~655 of 65 536 envs reset in every batched step).\n')
    log = os.path.join(src, 'stats_%s.log' % run)
    if os.path.exists(log):
        out += ['```'] + [l.rstrip() for l in open(log) if 'horizon' in l] + ['```', '']
    out.append('| kernel | calls | total ns | average ns | % | min ns | max ns |')
    out.append('|---|---|---|---|---|---|---|')
    for r in csv.DictReader(open(f)):
        name = r['Name'] if len(r['Name']) < 90 else r['Name'][:87] + '...'
        out.append('| `%s` | %s | %s | %.1f | %s | %s | %s |' % (name, r['Calls'], r['TotalDurationNs'], float(r['AverageNs']),
                                                              r['Percentage'], r['MinNs'], r['MaxNs']))
    out.append('')
open(os.path.join(dst, tag + '_kernel_stats.md'), 'w').write('\n'.join(out) + '\n')


def counter(run, name):
    f = one('pmc_%s_%s/**/*counter_collection.csv' % (run, name))
    modes = {'step': (0, 1), 'rollout': (1, 2), 'calib': (1, 9)}[run]      # the reset launch stays in as element [0]
    rows = [r for r in csv.DictReader(open(f)) if kernel_mode(r['Kernel_Name']) in modes and r['Counter_Name'] == name]
    return [float(r['Counter_Value']) for r in rows]


pm = ['# HBM traffic counters (%s)\n' % tag,
      'Separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (+ `--kernel-trace`), values in KiB per dispatch.\n']
# staging-only launch, per env: reads map 100 + inventory 36 + agent_location 8 + facing 4 + selected 1 + step_count 4 +
# episode 4 = 157 B; writes map 100 + inventory 36 + location 8 + facing 4 + reward 4 + done 1 + info 4 + selected 1 +
# step_count 4 + episode 4 = 166 B  (S = 10, K = 9; tools/dbg_launch.py calib)
N_CALIB = 1 << 20
rd_known, wr_known = N_CALIB * 157, N_CALIB * 166
cf, cw = counter('calib', 'FETCH_SIZE')[1:], counter('calib', 'WRITE_SIZE')[1:]       # [0] is the reset launch
f_factor = rd_known / (st.median(cf) * 1024)
w_factor = wr_known / (st.median(cw) * 1024)
pm += ['## Calibration on a known byte count\n',
       'Staging-only diagnostic kernel (`ngw_debug_launch` mode 9) at 1 048 576 envs: every launch reads %d B and writes %d B '
       '(state beyond the 256 MiB Infinity Cache).' % (rd_known, wr_known),
       '', '| counter | median KiB / launch | bytes | known bytes | known / counter |', '|---|---|---|---|---|',
       '| FETCH_SIZE | %.0f | %.0f | %d | %.3f |' % (st.median(cf), st.median(cf) * 1024, rd_known, f_factor),
       '| WRITE_SIZE | %.0f | %.0f | %d | %.3f |' % (st.median(cw), st.median(cw) * 1024, wr_known, w_factor), '',
       'gfx950 correction applied below: FETCH_SIZE x %.2f, WRITE_SIZE x %.2f (the guide says 2x / 1x for 16-B-per-lane streams).\n'
       % (round(f_factor), round(w_factor))]
fc, wc = round(f_factor), round(w_factor)
traffic = {}
pm += ['## Workload C2 (65 536 envs)\n', '| mode | launches | FETCH KiB | WRITE KiB | corrected HBM bytes / launch | algorithmic bytes / launch | ratio |',
       '|---|---|---|---|---|---|---|']
for mode, steps_per_launch in (('step', 1), ('rollout', 60)):
    f, w = counter(mode, 'FETCH_SIZE'), counter(mode, 'WRITE_SIZE')
    if mode == 'step':
        f, w = f[11:], w[11:]                          # skip reset + 10 warm-up launches
        fm, wm = st.mean(f), st.mean(w)
    else:
        fm, wm = f[-1], w[-1]                          # the timed 60-step rollout launch
    hbm = (fm * fc + wm * wc) * 1024
    alg = 353 * 65536 * steps_per_launch
    pm.append('| %s | %d | %.0f | %.0f | %.0f | %d | %.2f |' % (mode, len(f) if mode == 'step' else 1, fm, wm, hbm, alg, hbm / alg))
    traffic['C2_' + mode] = {'hbm_bytes_per_launch': round(hbm), 'env_steps_per_launch': 65536 * steps_per_launch,
                             'hbm_bytes_per_env_step': round(hbm / (65536 * steps_per_launch), 1),
                             'source': 'profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x%d)' % (tag, fc)}
pm += ['', 'The C2 working set (2 x 10 MB observation/state buffers) fits in the 256 MiB Infinity Cache, so these counters',
       'see only part of the algorithmic bytes (MI355X_MICROARCH.md §Infinity Cache: re-reads served on-die do not reach the',
       'memory-side counters); the calibration run above is the one that exercises HBM proper.']
open(os.path.join(dst, tag + '_pmc.md'), 'w').write('\n'.join(pm) + '\n')
json.dump(traffic, open(os.path.join(dst, 'pmc_traffic.json'), 'w'), indent=1)
print('\n'.join(out))
print('\n'.join(pm))
