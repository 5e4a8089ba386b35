"""Reset launch durations by mask pattern (kernel-trace timestamps): run through tools/reset_trace.sh.
Phases of 20 launches each: all envs / none (staging only) / one lane per wave / 8 lanes per wave."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
v = VecNovelGridworld(num_envs=n, seed=1)
L = _cabi.lib()
masks = [None, np.zeros(n, np.uint8), (np.arange(n) % 64 == 5).astype(np.uint8), (np.arange(n) % 8 == 3).astype(np.uint8),
         'sync', np.ones(n, np.uint8)]
for m in masks:
    for i in range(20):
        if isinstance(m, str):
            _cabi.check(L.ngw_reset(v._h, None)); v.sync()
        else:
            _cabi.check(L.ngw_reset(v._h, None if m is None else m.ctypes.data))
    v.sync()
