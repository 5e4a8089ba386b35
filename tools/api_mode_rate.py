"""PCIe-inclusive 'API mode' rate (never bench.py's `value`): host int32 actions in, full observation batch + outputs
copied back to host arrays on every step, through VecNovelGridworld.step()."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, reset_prefetch=0)
v.reset()
acts = np.random.randint(0, 17, size=(64, n)).astype(np.int32)
for i in range(10):
    v.step(acts[i])
t = time.perf_counter()
K = 100
for i in range(K):
    obs, reward, done, info = v.step(acts[i % 64])
dt = time.perf_counter() - t
print('API mode (H2D actions + D2H obs/outs every step): %d envs, %.1f us per batched step, %.1f M env-steps/s'
      % (n, dt / K * 1e6, n * K / dt / 1e6))
