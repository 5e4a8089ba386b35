"""Where a VecNovelGridworld.step() of a big batch spends its time (PCIe-inclusive host API, narrow wire format):
    make -C gym_novel_gridworlds_amd/csrc trace && NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_trace.so python tools/api_phases.py [n_envs ...]
The trace build prints, every 200 calls of ngw_step_host_packed, the host time of its three phases (actions validated + narrowed into the
page-locked buffer | launches enqueued | wait for the device); this script adds the wall time of the whole Python call and of the C call."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [65536, 16384]:
    v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100)
    v.reset()
    acts = np.random.RandomState(1).randint(0, 17, size=(64, n)).astype(np.int32)
    for i in range(20):
        v.step(acts[i])
    K = 400
    t = time.perf_counter()
    for i in range(K):
        v.step(acts[i % 64])
    py = (time.perf_counter() - t) / K * 1e6
    sys.stderr.flush()
    print('%d envs: VecNovelGridworld.step() %.1f us per call (Python + C call + device)' % (n, py), flush=True)
    v.close()
