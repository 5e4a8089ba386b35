"""In-kernel timeline of one all-envs reset launch of the dedicated new-episode kernel (diagnostics build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, apply_novelty, make_spec
NAMES = ['entry', 'LDS template + inventory rows', 'placement done', 'placed sorted, percent drawn', 'subset drawn', 'bit column final', 'map composed + stored', 'stores acked']
L = _cabi.lib()
L.ngw_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for wl in (sys.argv[1:] or ['C2', 'C5']):
    env_id, S, nov, n, desc = bench.WORKLOADS[wl]
    spec = make_spec(env_id, S)
    if nov:
        apply_novelty(spec, *nov)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=1, reset_prefetch=0)
    grid = (n + 63) // 64
    stamps = torch.zeros((grid, 32), dtype=torch.int64, device='cuda')
    _cabi.check(L.ngw_debug_set_stamps(v._h, C.c_void_p(stamps.data_ptr())))
    _cabi.check(L.ngw_reset(v._h, None)); v.sync()
    _cabi.check(L.ngw_reset(v._h, None)); v.sync()
    st = stamps.cpu().numpy()
    rt, cy = st[:, :8].astype(np.float64), st[:, 8:].astype(np.float64)
    t0 = rt[:, 0].min()
    print('== %s: %d waves' % (wl, grid))
    have = [i for i in range(8) if rt[:, i].max() > 0]            # (plain configurations have no subset stamps)
    for i in have:
        x = (rt[:, i] - t0) * 0.01
        print('%-32s min %9.2f  median %9.2f  max %9.2f us' % (NAMES[i], x.min(), np.median(x), x.max()))
    for i0, i1 in zip(have[:-1], have[1:]):
        d = cy[:, i1] - cy[:, i0]
        print('  %-32s -> %-32s %10.0f cycles (median)' % (NAMES[i0], NAMES[i1], np.median(d)))
    v.close()
