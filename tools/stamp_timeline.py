"""In-kernel timeline of ONE step launch (diagnostics build: make -C gym_novel_gridworlds_amd/csrc stamps).

    NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so python tools/stamp_timeline.py [C2 C4 ...]

Every wavefront records the chip-wide 100 MHz clock (s_memrealtime) and its shader clock (s_memtime) at: 0 entry,
1 all prologue loads issued, 2 data landed in LDS (after the barrier), 3 uniform scalars unpacked (loop entry),
4 step body done, 5 outputs issued, 6 every store acknowledged.  The launch sampled is the last of a back-to-back series
(hipGraph replay), i.e. steady state.  Prints, per stamp, when the waves reach it relative to the first wave's entry."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, apply_novelty, make_spec  # noqa: E402

NAMES = ['entry', 'loads issued', 'landed in LDS', 'scalars ready', 'step done', 'outputs issued', 'stores acked']
LEAN = ['entry', 'loads issued', 'landed in LDS', 'step begins', 'step done', 'outputs begin', 'outputs issued', 'stores acked']


def main():
    L = _cabi.lib()
    L.ngw_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    for wl in (sys.argv[1:] or ['C2']):
        env_id, S, nov, n, desc = bench.WORKLOADS[wl]
        n = int(os.environ.get('NGW_N', n))               # (NGW_N=1: the single-env adapter's handle, state in host memory)
        spec = make_spec(env_id, S)
        if nov:
            apply_novelty(spec, *nov)
        A = len(spec.actions_id)
        stagger = os.environ.get('NGW_STAGGER') == '1'   # episode ends spread over the batch, rows prepared: the cold path in the sampled launch
        v = VecNovelGridworld(spec=spec, num_envs=n, device=0, seed=0, autoreset=True, horizon=100, reset_prefetch=(1 << 20) if stagger else 0)   # (rows prepared by reset(); no refill launch among the sampled steps)
        v.reset()
        if os.environ.get('NGW_LIDAR') == '1':          # the fused LidarInFront epilogue sits between 'outputs begin' and 'outputs issued'
            v.lidar_configure(num_beams=8, fused=True, dtype={'int32': np.int32, 'int16': np.int16}.get(os.environ.get('NGW_LIDAR_DTYPE', 'int16'), 'packed'))
        if stagger:
            v.set_state(0, step_count=(np.arange(n) * 7919 % 100).astype(np.int32))
        grid = (n + 63) // 64
        stamps = torch.zeros((grid, 32), dtype=torch.int64, device='cuda')
        acts = torch.randint(0, A, (40, n), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        _cabi.check(L.ngw_debug_set_stamps(v._h, C.c_void_p(stamps.data_ptr())))
        if stagger:                                      # eager launches: a captured graph with prepared episodes ends in a refill launch
            for i in range(30):
                v.step_device(acts[i].data_ptr())
            v.sync()
            stamps.zero_()
            torch.cuda.synchronize()
            v.timing_begin()
            v.step_device(acts[30].data_ptr())
            ms = v.timing_end() * 40
        else:
            v.graph_build(acts.data_ptr(), n, 40)
            v.graph_launch(1)
            v.sync()
            v.timing_begin()
            v.graph_launch(1)
            ms = v.timing_end()
        st = stamps.cpu().numpy()
        rt, cy, sub = st[:, :8].astype(np.float64), st[:, 8:16].astype(np.float64), st[:, 16:].astype(np.float64)
        t0 = rt[:, 0].min()
        names = LEAN if rt[:, 7].max() > 0 else NAMES           # the lean kernel fills all eight slots
        print('== %s%s: %s; %d waves; event time per launch %.2f us' % (wl, ' (staggered episode ends)' if stagger else '', desc, grid, ms / 40 * 1e3))
        print('%-16s %8s %8s %8s %8s %8s   (us after the first wave entered)' % ('stamp', 'min', 'p10', 'median', 'p90', 'max'))
        for i, nm in enumerate(names):
            x = (rt[:, i] - t0) * 0.01
            print('%-16s %8.2f %8.2f %8.2f %8.2f %8.2f' % (nm, x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max()))
        print('per-wave shader cycles between stamps (median / p90):')
        last = len(names) - 1
        for i in range(1, last + 1):
            d = cy[:, i] - cy[:, i - 1]
            print('  %-16s -> %-16s %8.0f %8.0f' % (names[i - 1], names[i], np.median(d), np.percentile(d, 90)))
        if stagger and last >= 6:                        # the cold path sits between 'outputs begin' and 'outputs issued'
            d = cy[:, 6] - cy[:, 5]
            print('  outputs begin -> outputs issued, percentiles 10/25/50/75/90/99: ' + ' '.join('%.0f' % np.percentile(d, q) for q in (10, 25, 50, 75, 90, 99)))
        if sub.max() > 0:                                # stamps inside the LidarInFront row builder (STAMP_SUB in lidar_rows)
            sn = ['epilogue entered', 'tile zeroed', 'cells read', 'hits written', 'inventory written', 'rows stored (issued)', 'rays resolved']
            order = [0, 1, 2, 6, 3, 4, 5]                  # (slot 6 sits between 'cells read' and 'hits written')
            if v.step_reads_map_in_place:                  # the bit-row form (ngw_boards.inc): its own stations
                sn = ['epilogue entered', 'lines cut out of the bit rows', 'rays resolved, hit cells requested, inventory tail written', 'hits written', '-', 'rows stored (issued)', 'hit cells landed']
                order = [0, 1, 2, 6, 3, 5]
            have = [i for i in order if sub[:, i].max() > 0]
            print('  inside the lidar epilogue, shader cycles (median / p90):')
            for a_, b_ in zip(have[:-1], have[1:]):
                d = sub[:, b_] - sub[:, a_]
                print('    %-20s -> %-22s %8.0f %8.0f' % (sn[a_], sn[b_], np.median(d), np.percentile(d, 90)))
        life = cy[:, last] - cy[:, 0]
        ghz = float(np.median(life / np.maximum((rt[:, last] - rt[:, 0]) * 10.0, 1.0)))
        # what a stamp itself costs (two clock reads + the wait for them): the shortest median distance between two adjacent stamps - two of
        # them sit back to back with nothing in between - times the stamps a wave passes after its first
        stamp_cost = float(min(np.median(cy[:, i] - cy[:, i - 1]) for i in range(1, last + 1)))
        net = float(np.median(life)) - last * stamp_cost
        print('  wave life %.0f cycles median (%.0f net of the %d stamps behind the first, %.0f cycles each); clock ~%.2f GHz' % (np.median(life), net, last, stamp_cost, ghz))
        if os.environ.get('NGW_STAMP_JSON'):              # bench.py's roofline.floor reads the median wave life from profiles/pmc_traffic.json (tools/parse_round.py merges this file)
            import json
            life_us = (rt[:, last] - rt[:, 0]) * 0.01       # the chip-wide 100 MHz clock: 10 ns resolution per wave, the median over 1 024 waves is what counts
            rec = {'median_cycles': float(np.median(life)), 'clock_ghz': round(ghz, 3), 'median_us': round(float(np.median(life)) / (ghz * 1e3), 3),
                   'stamp_cost_cycles': stamp_cost, 'stamps_after_first': int(last), 'median_us_net_of_stamps': round(net / (ghz * 1e3), 3),
                   'median_us_realtime_clock': round(float(np.median(life_us)), 3), 'last_wave_done_us': round(float((rt[:, last].max() - t0) * 0.01), 3),
                   'event_us_per_launch': round(ms / 40 * 1e3, 3), 'waves': int(grid)}
            path = os.environ['NGW_STAMP_JSON']
            allrec = json.load(open(path)) if os.path.exists(path) else {}
            allrec[wl] = rec
            json.dump(allrec, open(path, 'w'), indent=1)
        _cabi.check(L.ngw_debug_set_stamps(v._h, None))
        v.close()


if __name__ == '__main__':
    main()
