"""Duration of a reset launch (HIP event pair on the handle's stream): all envs, one lane per wave, and the prepared-
episode refill after all envs consumed theirs.  NGW_FAST_RESET=0 times the general kernel (A/B)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, apply_novelty, make_spec

L = _cabi.lib()
for wl in (sys.argv[1:] or ['C2', 'C3', 'C4', 'C5']):
    env_id, S, nov, n, desc = bench.WORKLOADS[wl]
    spec = make_spec(env_id, S)
    if nov:
        apply_novelty(spec, *nov)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=1, autoreset=True, horizon=100, reset_prefetch=0)
    one = (np.arange(n) % 64 == 5).astype(np.uint8)
    out = []
    for name, m, reps in (('all envs', None, 5), ('one lane per wave', one, 5)):
        _cabi.check(L.ngw_reset(v._h, None if m is None else m.ctypes.data)); v.sync()
        v.timing_begin()
        for _ in range(reps):
            _cabi.check(L.ngw_reset(v._h, None if m is None else m.ctypes.data))
        out.append('%s %.1f us' % (name, v.timing_end() / reps * 1e3))
    v.set_reset_prefetch(1 << 20)                       # refills only after explicit resets
    ms = []
    for _ in range(3):
        v.timing_begin()
        _cabi.check(L.ngw_reset(v._h, None))            # reset launch + refill launch of every row
        ms.append(v.timing_end() * 1e3)
    out.append('reset + refill of all rows %.1f us' % min(ms))
    assert v.error_flags() == 0
    print('%s fast_reset=%s: %s' % (wl, os.environ.get('NGW_FAST_RESET', '1'), '; '.join(out)), flush=True)
    v.close()
