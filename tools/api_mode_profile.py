"""cProfile of the host-side API loop (VecNovelGridworld.step with host actions and host observations)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld  # noqa: E402

n = 65536
v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, reset_prefetch=0)
v.reset()
acts = np.random.randint(0, 17, size=(64, n)).astype(np.int32)
for i in range(10):
    v.step(acts[i])
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    v.step(acts[i % 64])
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
