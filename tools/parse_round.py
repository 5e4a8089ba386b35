#!/usr/bin/env python3
"""tools/parse_round.py <round>: turns gpurun_out/prof_<round>/ (written by tools/profile_round.sh <round>) into the committed summaries:
   profiles/<round>_kernel_stats.md   rocprofv3 --kernel-trace --stats of the bench command (the driver's own, every workload, the fused rollout)
   profiles/<round>_pmc.md            HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) with the calibration on a
                                  known byte count, the new-episode kernel's write traffic against its payload, and the SQ
                                  counters of the C2 step kernel and of the fused rollout
   profiles/pmc_traffic.json      per-launch HBM bytes that bench.py reports as roofline.traffic (a constant of the kernel)
Counter values are KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reads wide coalesced reads by 2x; the
factor is re-measured here on the staging-only diagnostic kernel whose byte count is known exactly."""
import collections
import csv
import glob
import json
import os
import re
import statistics as st
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r04'

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree_commit():
    """HEAD of the tree being parsed, taken BEFORE this script writes anything: '-dirty' only if a file outside profiles/ (this script's
    own output directory) differs from HEAD.  The GPU box gets a snapshot without .git, so the tree gpurun sent is the one parsed here."""
    import subprocess
    try:
        head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        dirt = subprocess.run(['git', 'status', '--porcelain', '--', '.', ':(exclude)profiles'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        return 'unknown'
    return (head or 'unknown') + ('-dirty' if dirt else '') + ' (tree at parse time)'


TREE_COMMIT = _tree_commit()
SRC = os.path.join(ROOT, 'gpurun_out', 'prof_' + ROUND)
DST = os.path.join(ROOT, 'profiles')
WL = {'C2': (65536, 353, 'Pogostick-v1 10x10, 65 536 envs'), 'C3': (65536, 953, 'Bow-v1 20x20, 65 536 envs'),
      'C4': (32768, 365, 'Pogostick-v1 + axe(medium) 10x10, 32 768 envs'), 'C5': (65536, 2213, 'Pogostick-v1 + additem(hard) 32x32, 65 536 envs')}


def first(pattern):
    """The NEWEST match: gpurun merges a call's files into the local gpurun_out/ without removing older ones, so a directory
    may hold the CSVs of several profiling runs (rocprofv3 names them by process id)."""
    f = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def short(name):
    m = re.search(r'(ngw_\w+<[^>]*>)', name)
    return m.group(1) if m else (name[:60] + '...' if len(name) > 60 else name)


def bench_line(log):
    try:
        return json.loads([ln for ln in open(os.path.join(SRC, log)) if ln.startswith('{')][-1])
    except (OSError, IndexError, ValueError):
        return None


# ---------------------------------------------------------------- kernel stats
out = ['# rocprofv3 --kernel-trace --stats summaries (%s)\n' % ROUND,
       'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-side --steps 400 --warmup 100 '
       '--workload <W> [--mode rollout]`, and the driver\'s own `python3 bench.py --gpus 1 --steps 20 --warmup 5` (`tools/profile_round.sh %s`; raw CSVs under `gpurun_out/prof_%s/`).\n' % (ROUND, ROUND),
       '**How to read the step-kernel durations.**  The step kernels of this round run 2.0-2.6 us of wave activity (in-kernel clock '
       'stamps, `tools/stamp_timeline.py`) inside a 3.7-4.9 us launch period when the launches are replayed back to back from a '
       'hipGraph (what `bench.py` times with a HIP event pair: the command processor prepares dispatch i+1 while dispatch i runs).  Under '
       '`rocprofv3 --kernel-trace` every dispatch is bracketed by profiling signals and its start stamp is taken when the packet is picked '
       'up, so a kernel this short shows its whole un-overlapped dispatch: the same bench command reports %s us per batched step while it '
       'is being traced (its own JSON line below) against 3.7-3.9 us untraced, and the per-kernel average of the trace sits '
       'between the two.  Both numbers are given; `roofline.achieved` uses the untraced event-pair average, as the bench contract says.\n',
       '**The ~1 600 `ngw_rollout_lean` + refill (`ngw_kernel<., 3, .>`) launches at the top of every table** are `bench.py`\'s 250 ms device clock '
       'warm-up on a SCRATCH handle (`--clock-warm-ms`), before anything is measured; the measured handle\'s launches are the `ngw_step_lean` rows '
       '(warm-up + timed steps) and the few reset / refill launches beside them.\n']
ms = []
for W in ('C2', 'C3', 'C4', 'C5'):
    ln = bench_line('stats_%s.log' % W)
    if ln:
        ms.append('%s %.1f' % (W, ln['ms_per_step'] * 1e3))
out[-2] = out[-2] % ', '.join(ms)                        # (the paragraph about traced vs untraced step times)
for W, mode in (('driver', 'step'), ('C2', 'step'), ('C3', 'step'), ('C4', 'step'), ('C5', 'step'), ('C2_rollout', 'rollout')):
    f = first('stats_%s/**/*kernel_stats.csv' % W)
    if not f:
        continue
    base = 'C2' if W == 'driver' else W.split('_')[0]
    out.append('## %s (%s), %s mode - kernel_stats.csv\n' % ("the driver's command: 20 steps (one replay of a 20-node graph per region since the end of round 5), 5 warm-up" if W == 'driver' else W, WL[base][2], mode))
    ln = bench_line('stats_%s.log' % W)
    if ln:
        out.append('bench line of the traced run: %.2f G env-steps/s, %.2f us per batched step, %d resets of every env in the timed region\n'
                   % (ln['value'] / 1e9, ln['ms_per_step'] * 1e3, ln.get('resets_in_timed_region', -1)))
    out.append('| kernel | calls | total ns | average ns | % | min ns | max ns |')
    out.append('|---|---|---|---|---|---|---|')
    for r in csv.DictReader(open(f)):
        if 'ngw' not in r['Name'] and float(r['Percentage']) < 1.0:
            continue
        out.append('| `%s` | %s | %s | %.1f | %s | %s | %s |' % (short(r['Name']), r['Calls'], r['TotalDurationNs'], float(r['AverageNs']),
                                                              r['Percentage'], r['MinNs'], r['MaxNs']))
    tr = first('stats_%s/**/*kernel_trace.csv' % W)
    by = collections.defaultdict(list)
    rows = list(csv.DictReader(open(tr)))
    for r in rows:
        if 'ngw' in r['Kernel_Name']:
            by[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out.append('')
    out.append('kernel trace, per kernel: ' + '; '.join('`%s` %d launches, median %.2f us, mean %.2f us, max %.1f us' % (k, len(v), st.median(v), st.mean(v), max(v))
                                                        for k, v in by.items()))
    if mode == 'step':
        ks = [r for r in rows if 'ngw_step_lean' in r['Kernel_Name']]
        if len(ks) > 200:
            s = [int(r['Start_Timestamp']) for r in ks]
            per = [(s[i + 1] - s[i]) / 1e3 for i in range(100, len(s) - 1)]
            out.append('\nstart-to-start period of the step launches inside the replayed graph (traced): median %.2f us' % st.median(per))
    out.append('')
open(os.path.join(DST, ROUND + '_kernel_stats.md'), 'w').write('\n'.join(out) + '\n')


# ---------------------------------------------------------------- PMC
def counters(d):
    f = first('%s/**/*counter_collection.csv' % d)
    by = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if 'ngw' in r['Kernel_Name']:
                by[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
    return by


pm = ['# HBM traffic and SQ counters (%s)\n' % ROUND,
      'rocprofv3 `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes (the TCC slots do not fit both), `--kernel-trace` only beside them; '
      'command `python3 bench.py --no-cpu-baseline --no-side --steps 60 --warmup 10 --launch eager --workload <W>` (`tools/profile_round.sh`).  Values are per launch, '
      'median over the 70 step launches of a pass.\n']
cal = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    by = counters('pmc_calib_%s' % c)
    for (k, cn), v in by.items():
        if ', 9,' in k:
            cal[c] = st.median(v)
n_cal = 1 << 20
ff = n_cal * 157 / (cal['FETCH_SIZE'] * 1024) if 'FETCH_SIZE' in cal else 2.0
wf = n_cal * 166 / (cal['WRITE_SIZE'] * 1024) if 'WRITE_SIZE' in cal else 1.0
pm.append('## Calibration on a known byte count\n')
pm.append('Staging-only diagnostic kernel (`ngw_debug_launch` mode 9) at 1 048 576 envs (state far beyond the 256 MiB Infinity Cache): every launch reads '
          '%d B and writes %d B.  FETCH_SIZE reported %.0f KiB -> factor **%.3f** (the guide\'s x2 for wide coalesced reads on gfx950), WRITE_SIZE %.0f KiB -> factor **%.3f**.  '
          'The step kernels read their few map cells with byte loads, an access width the guide calls uncalibrated: their corrected read figure is an estimate '
          '(L2 fills are 128-B lines either way; the expected line traffic, ~1.5 map lines + 57 B of scalars and inventory per env, agrees with it within 15 %%).\n'
          % (n_cal * 157, n_cal * 166, cal.get('FETCH_SIZE', 0), ff, cal.get('WRITE_SIZE', 0), wf))
pm.append('## Step kernels\n')
pm.append('| workload | kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes per launch (corrected) | per env-step | SURVEY §8(d) algorithmic bytes per env-step |')
pm.append('|---|---|---|---|---|---|---|')
traffic = {}
for W in ('C2', 'C3', 'C4', 'C5'):
    f_by, w_by = counters('pmc_%s_FETCH_SIZE' % W), counters('pmc_%s_WRITE_SIZE' % W)
    for (k, cn), v in f_by.items():
        if 'step_lean' not in k:
            continue
        fv, wv = st.median(v), st.median(w_by.get((k, 'WRITE_SIZE'), [0]))
        total = fv * 1024 * ff + wv * 1024 * wf
        n = WL[W][0]
        pm.append('| %s | `%s` | %.1f | %.1f | %.0f | %.1f | %d |' % (W, k, fv, wv, total, total / n, WL[W][1]))
        traffic['%s_step' % W] = {'hbm_bytes_per_launch': round(total), 'env_steps_per_launch': n, 'hbm_bytes_per_env_step': round(total / n, 1),
                                  'source': 'profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x%.3f)' % (ROUND, ff)}
pm.append('\nThe step kernels move well under the 2*S*S + 12*K + 45 bytes the survey prices for a read-pack-write design: the observation buffers are the '
          'state, updated in place, and a step reads only the map lines around the agent (since round 4 at every map size: `ngw_step_lean<., false, ., .>`).  '
          '`bench.py` prices `roofline.achieved` / `frac` on the survey figure for every workload (the figure all rounds are compared on - for the big maps it '
          'exceeds 1, which says how the step compares with a read-pack-write design at the peak, not how busy HBM is), reports the bytes this design has to move '
          'as a separate, labelled field (`design_bytes_per_env_step`) and divides the bytes of THIS table by the launch time in `frac_of_peak_on_measured_traffic`.\n')
pm.append('## Other launches of the same passes\n')
for W in ('C2', 'C3', 'C4', 'C5'):
    f_by, w_by = counters('pmc_%s_FETCH_SIZE' % W), counters('pmc_%s_WRITE_SIZE' % W)
    for (k, cn), v in f_by.items():
        if 'step_lean' in k:
            continue
        wv = w_by.get((k, 'WRITE_SIZE'), [0])
        pm.append('* %s `%s` (%d launches): FETCH_SIZE max %.0f KiB, WRITE_SIZE max %.0f KiB' % (W, k, len(v), max(v), max(wv)))
pm.append('')
# rollout traffic
f_by, w_by = counters('pmc_C2rollout_FETCH_SIZE'), counters('pmc_C2rollout_WRITE_SIZE')
for (k, cn), v in f_by.items():
    if 'rollout_lean' in k or ', 2,' in k:
        fv, wv = max(v), max(w_by.get((k, 'WRITE_SIZE'), [0]))
        total = fv * 1024 * ff + wv * 1024 * wf
        steps = 60
        pm.append('## Fused rollout (C2, %d steps per launch)\n' % steps)
        pm.append('`%s`: FETCH_SIZE %.1f KiB, WRITE_SIZE %.1f KiB per launch -> %.0f B = **%.1f B per env-step**: every step overwrites the same lines in L2, only the '
                  'last values leave the chip.  The fused rollout is bound by instruction issue, not by HBM (SQ counters below).\n' % (k, fv, wv, total, total / (65536 * steps)))
        traffic['C2_rollout'] = {'hbm_bytes_per_launch': round(total), 'env_steps_per_launch': 65536 * steps, 'hbm_bytes_per_env_step': round(total / (65536 * steps), 1),
                                 'source': 'profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x%.3f)' % (ROUND, ff)}
# new-episode kernel: write traffic against its payload
RS = {'C3': (65536, 20, 9, 'Bow-v1 20x20'), 'C5': (65536, 32, 10, 'Pogostick-v1 + additem(hard) 32x32'), 'X1': (65536, 10, 10, 'Pogostick-v1 + firewall(hard) 10x10')}
rows = []
for W, (n, S, K, what) in RS.items():
    f_by, w_by = counters('pmc_reset_%s_FETCH_SIZE' % W), counters('pmc_reset_%s_WRITE_SIZE' % W)
    for (k, cn), v in w_by.items():
        if 'reset_fast' not in k and ', 1,' not in k:
            continue
        wv, fv = st.median(v), st.median(f_by.get((k, 'FETCH_SIZE'), [0]))
        payload = n * (S * S + 4 * K + 16)
        rows.append('| %s (%s) | `%s` | %.0f | %.0f | %.1f | %.1f | **%.2f** |' % (W, what, k, fv, wv, wv * 1024 * wf / 1e6, payload / 1e6, wv * 1024 * wf / payload))
        traffic['%s_reset' % W] = {'write_bytes_per_launch': round(wv * 1024 * wf), 'payload_bytes_per_launch': payload, 'write_amplification': round(wv * 1024 * wf / payload, 3),
                                   'source': 'profiles/%s_pmc.md (tools/reset_pmc.py, rocprofv3 --pmc WRITE_SIZE)' % ROUND}
if rows:
    pm.append('## New-episode kernel: HBM writes against the payload\n')
    pm.append('`python3 tools/reset_pmc.py <W>`: six explicit resets of every env, prepared episodes off (every launch builds its episodes).  Payload of a launch = '
              'N x (S*S + 4*K + 16) bytes: the map, the inventory row, pose / facing / counters.  Round 2 wrote the ring-and-air template first and scattered the items '
              'as single bytes afterwards: 794 MB per launch at C5 for 70.8 MB of payload (11.2x), 46.6 MB at C3 for 29.6 MB (1.6x).  The rows are now composed per env '
              'in an LDS tile and stored once, with consecutive lanes on consecutive bytes of one env.\n')
    pm.append('| workload | kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | written MB | payload MB | written / payload |')
    pm.append('|---|---|---|---|---|---|---|')
    pm += rows
    pm.append('')
# SQ
pm.append('## SQ counters\n')
pm.append('Two `--pmc` passes of 7 / 6 SQ counters each (the SQ block has 8 slots).  `SQ_WAVE_CYCLES`, `SQ_WAIT_*`, `SQ_ACTIVE_INST_*` count quad-cycles.\n')
for label, dirs, pick, per in (('C2 step kernel `ngw_step_lean<0, false, false, false>` (per launch: 1024 waves, one batched step)', ('sq_step_1', 'sq_step_2'), 'step_lean', 1),
                               ('C2 fused rollout `ngw_rollout_lean<0, false, false, false>` (200 steps per launch, prepared episodes off)', ('sq_rollout_1', 'sq_rollout_2'), 'rollout_lean', 200)):
    vals = {}
    for d in dirs:
        for (k, cn), v in counters(d).items():
            if pick in k:
                vals[cn] = st.median(v) if per == 1 else max(v)
    if not vals:
        continue
    pm.append('### %s\n' % label)
    pm.append('| counter | per launch | per wave%s |' % (' and step' if per > 1 else ''))
    pm.append('|---|---|---|')
    waves = vals.get('SQ_WAVES', 1024) or 1024
    for cn in sorted(vals):
        pm.append('| %s | %.0f | %.1f |' % (cn, vals[cn], vals[cn] / waves / per))
    if 'SQ_WAVE_CYCLES' in vals:
        wc = vals['SQ_WAVE_CYCLES']
        pm.append('\nwaiting (SQ_WAIT_ANY) %.0f %% of the wave cycles, issuing (SQ_ACTIVE_INST_ANY) %.0f %%, issue stalls (SQ_WAIT_INST_ANY) %.0f %%\n'
                  % (100 * vals.get('SQ_WAIT_ANY', 0) / wc, 100 * vals.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * vals.get('SQ_WAIT_INST_ANY', 0) / wc))
    if per > 1:
        insts = sum(vals.get(c, 0) for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM', 'SQ_INSTS_BRANCH'))
        traffic['C2_rollout_sq'] = {'instructions_per_wave_step': round(insts / waves / per, 1),
                                    'valu': round(vals.get('SQ_INSTS_VALU', 0) / waves / per, 1), 'salu': round(vals.get('SQ_INSTS_SALU', 0) / waves / per, 1),
                                    'branch': round(vals.get('SQ_INSTS_BRANCH', 0) / waves / per, 1), 'lds': round(vals.get('SQ_INSTS_LDS', 0) / waves / per, 1),
                                    'issue_share_of_wave_cycles': round(vals.get('SQ_ACTIVE_INST_ANY', 0) / vals.get('SQ_WAVE_CYCLES', 1), 3),
                                    'source': 'profiles/%s_pmc.md (rocprofv3 --pmc SQ_*, 200-step launch)' % ROUND}
open(os.path.join(DST, ROUND + '_pmc.md'), 'w').write('\n'.join(pm) + '\n')

# ---------------------------------------------------------------- the fused LidarInFront step
ld = ['# The step with the fused LidarInFront observation (%s)\n' % ROUND,
      'Reference: `gym_novel_gridworlds/observation_wrappers.py:10-80` - the observation every reference training / evaluation script wraps the env in.  '
      'Kernel `ngw_step_lean<0, false, EXT, true, NR>` (round 5): the IN-PLACE step kernel with the observation built from the occupancy bit rows '
      '(`csrc/ngw_boards.inc`: the first hit of each of the 8 rays is a count-leading / trailing-zeros on the agent\'s row, column and two diagonals; NR = 12 / 20 / 32 '
      'register rows for maps up to 12 / 20 / 32 cells), the rows (8 beams x 7 lidar items + the inventory tail) written in the same launch.  Commands: '
      '`python3 bench.py --no-cpu-baseline --no-side [--workload W] --lidar <format> ...` (`tools/profile_round.sh`, section 5); the default bench line reports the three '
      'formats at C2 untraced under `lidar`.\n']
ld.append('## Kernel trace per workload and row format (400 timed steps, hipGraph replay)\n')
ld.append('| workload | rows | bytes per env | bench line while traced: us per batched step | `ngw_step_lean<..., true, NR>` launches | average ns (kernel_stats.csv) | median us (kernel trace) |')
ld.append('|---|---|---|---|---|---|---|')
for W, F in (('C2', 'int32'), ('C2', 'int16'), ('C2', 'packed'), ('C3', 'packed'), ('C5', 'packed')):
    tag = F if W == 'C2' else '%s_%s' % (W, F)
    f = first('stats_lidar_%s/**/*kernel_stats.csv' % tag)
    ln = bench_line('stats_lidar_%s.log' % tag)
    if not f or not ln:
        continue
    row = [r for r in csv.DictReader(open(f)) if 'step_lean' in r['Name']]
    tr = first('stats_lidar_%s/**/*kernel_trace.csv' % tag)
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(tr)) if 'step_lean' in r['Kernel_Name']]
    if row:
        ld.append('| %s | %s | %d | %.2f | %s | %.1f | %.2f |' % (W, F, ln['config']['fused_lidar']['row_bytes'], ln['ms_per_step'] * 1e3, row[0]['Calls'], float(row[0]['AverageNs']),
                                                            st.median(dur) if dur else float('nan')))
ld.append('\n(As for the plain step kernel, a traced dispatch shows its whole un-overlapped launch; the untraced figures are the bench line\'s `lidar` key and `tools/lidar_rate.py`.)\n')
ld.append('## HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes)\n')
ld.append('| workload | rows | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes per launch (corrected) | per env-step | of which observation rows |')
ld.append('|---|---|---|---|---|---|---|')
for W, F in (('C2', 'int16'), ('C2', 'packed'), ('C3', 'packed'), ('C5', 'packed')):
    tag = F if W == 'C2' else '%s_%s' % (W, F)
    f_by, w_by = counters('pmc_lidar_%s_FETCH_SIZE' % tag), counters('pmc_lidar_%s_WRITE_SIZE' % tag)
    ln = bench_line('pmc_lidar_%s_FETCH_SIZE.log' % tag)
    for (k, cn), v in f_by.items():
        if 'step_lean' not in k:
            continue
        fv, wv = st.median(v), st.median(w_by.get((k, 'WRITE_SIZE'), [0]))
        total = fv * 1024 * ff + wv * 1024 * wf
        rb = ln['config']['fused_lidar']['row_bytes'] if ln else 0
        envs = ln['config']['envs_per_gpu'] if ln else 65536
        ld.append('| %s | %s | %.1f | %.1f | %.0f | %.1f | %d |' % (W, F, fv, wv, total, total / envs, rb))
        traffic['%s_lidar_%s_step' % (W, F)] = {'hbm_bytes_per_launch': round(total), 'env_steps_per_launch': envs, 'hbm_bytes_per_env_step': round(total / envs, 1),
                                                'source': 'profiles/%s_lidar.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)' % ROUND}
ld.append('\n## SQ counters (int16 rows)\n')
vals = {}
for d in ('sq_lidar_1', 'sq_lidar_2', 'sq_lidar_3'):
    for (k, cn), v in counters(d).items():
        if 'step_lean' in k:
            vals[cn] = st.median(v)
if vals:
    waves = vals.get('SQ_WAVES', 1024) or 1024
    ld.append('| counter | per launch | per wave |')
    ld.append('|---|---|---|')
    for cn in sorted(vals):
        ld.append('| %s | %.0f | %.1f |' % (cn, vals[cn], vals[cn] / waves))
    if 'SQ_WAVE_CYCLES' in vals:
        wc = vals['SQ_WAVE_CYCLES']
        ld.append('\nwaiting (SQ_WAIT_ANY) %.0f %% of the wave cycles, issuing (SQ_ACTIVE_INST_ANY) %.0f %%\n' % (100 * vals.get('SQ_WAIT_ANY', 0) / wc, 100 * vals.get('SQ_ACTIVE_INST_ANY', 0) / wc))
for F in ('int16', 'packed', 'int32'):                    # the stamped timelines of tools/lidar_probe.sh, if that ran in this round
    lg = os.path.join(ROOT, 'gpurun_out', ROUND, 'stamp_lidar_%s.log' % F)
    if os.path.exists(lg):
        ld.append('## In-kernel timeline, %s rows (`tools/lidar_probe.sh`: diagnostics build with clock stamps, C2, last launch of a replayed graph)\n' % F)
        ld.append('```')
        ld += [x.rstrip() for x in open(lg) if 'amdgpu.ids' not in x]
        ld.append('```\n')
ll = os.path.join(ROOT, 'gpurun_out', ROUND, 'lds_layout.log')
if os.path.exists(ll):
    ld.append('## LDS per wavefront (`NGW_DEBUG_LDS=1`, packed rows; 160 KB per CU, one wavefront per workgroup)\n')
    ld.append('```')
    ld += [x.rstrip() for x in open(ll) if x.startswith('[ngw]') or x.startswith('==')]
    ld.append('```\n')
for W in ('', '_C2', '_C3', '_C5'):
    lr = os.path.join(ROOT, 'gpurun_out', ROUND, 'lidar_rate%s.log' % W)
    if os.path.exists(lr):
        ld.append('## Untraced rates%s (`tools/lidar_rate.py`: 64-step hipGraph replayed 16 times, default prepared-episode cadence)\n' % (W and ', ' + W[1:]))
        ld.append('```')
        ld += [x.rstrip() for x in open(lr) if 'amdgpu.ids' not in x]
        ld.append('```\n')
open(os.path.join(DST, ROUND + '_lidar.md'), 'w').write('\n'.join(ld) + '\n')


def _read(name):
    try:
        return open(os.path.join(SRC, name)).read().strip()
    except OSError:
        return 'unknown'


# median life of a wave of the step kernel (in-kernel clock stamps, tools/stamp_timeline.py): what bench.py's roofline.floor adds to the
# empty-kernel launch period it measures itself
wl_file = os.path.join(SRC, 'wave_life.json')
if os.path.exists(wl_file):
    lg0 = os.path.join(SRC, 'stamps_step.log')
    gaps = {}                                             # (logs of a run whose JSON has no stamp cost yet: the per-stamp cost from the printed medians)
    if os.path.exists(lg0):
        cur = None
        for line in open(lg0):
            if line.startswith('== '):
                cur = line[3:].split(':')[0].split(' ')[0]
            elif '->' in line and cur and 'lidar' not in line:
                try:
                    gaps.setdefault(cur, []).append(float(line.split()[-2]))
                except ValueError:
                    pass
    for wl_name, rec in json.load(open(wl_file)).items():
        if 'median_us_net_of_stamps' not in rec and gaps.get(wl_name):
            g = gaps[wl_name][:7]
            rec['stamp_cost_cycles'], rec['stamps_after_first'] = min(g), len(g)
            rec['median_us_net_of_stamps'] = round((rec['median_cycles'] - len(g) * min(g)) / (rec['clock_ghz'] * 1e3), 3)
        rec['source'] = 'profiles/%s_kernel_stats.md (tools/stamp_timeline.py on the -DNGW_STAMPS build: s_memtime per wave, median over %d waves)' % (ROUND, rec.get('waves', 0))
        traffic['%s_wave_life' % wl_name] = rec
    ks_path = os.path.join(DST, ROUND + '_kernel_stats.md')
    lg = os.path.join(SRC, 'stamps_step.log')
    if os.path.exists(lg) and os.path.exists(ks_path):
        with open(ks_path, 'a') as f:
            f.write('\n## In-kernel timeline of the step kernel (`tools/stamp_timeline.py`, diagnostics build with clock stamps; last launch of a replayed 40-step graph)\n\n```\n')
            f.write(''.join(x for x in open(lg) if 'amdgpu.ids' not in x))
            f.write('```\n')

commit = _read('commit.txt')
if commit == 'unknown':                                  # (no .git on the GPU box)
    commit = TREE_COMMIT
traffic['_provenance'] = {'round': ROUND, 'commit': commit, 'measured': _read('date.txt'),
                          'how': 'tools/profile_round.sh %s on one MI355X (gpurun), parsed by tools/parse_round.py' % ROUND}
json.dump(traffic, open(os.path.join(DST, 'pmc_traffic.json'), 'w'), indent=1)
print('wrote profiles/%s_kernel_stats.md, profiles/%s_pmc.md, profiles/pmc_traffic.json' % (ROUND, ROUND))
print(json.dumps(traffic, indent=1))
