"""A/B of terminal-observation capture (ngw_set_terminal_capture) on the per-launch step at C2, 65 536 envs, H = 100, prepared next
episodes at the default cadence, hipGraph replay of 150 steps (two refill periods):
    python tools/terminal_ab.py            -> one line per (regime, capture) pair, alternated REPS times
regimes: 'synchronised' - every env ends its episode in the same batched step (the cold path runs in 1 step of 100);
         'staggered'    - step_count offset e * 7919 % H, ~655 envs end per batched step (the cold path runs in every step)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_novel_gridworlds_amd import VecNovelGridworld, make_spec  # noqa: E402

n, H, G = 65536, 100, 150
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spec = make_spec('NovelGridworld-Pogostick-v1', 10)
A = len(spec.actions_id)
acts = torch.randint(0, A, (G, n), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()

# clock warm-up on a scratch handle
s = VecNovelGridworld(spec=spec, num_envs=n, autoreset=True, horizon=H); s.reset()
t_end = time.perf_counter() + 0.25
while time.perf_counter() < t_end:
    s.rollout(200, 1, 0); s.sync()
s.close()


def run(staggered, capture):
    v = VecNovelGridworld(spec=spec, num_envs=n, autoreset=True, horizon=H, terminal_capture=capture)
    v.reset()
    if staggered:
        v.set_state(0, step_count=(np.arange(n) * 7919 % H).astype(np.int32))
    v.graph_build(acts.data_ptr(), n, G)
    v.graph_launch(2)
    v.sync()
    v.timing_begin()
    v.graph_launch(8)
    ms = v.timing_end()
    assert v.error_flags() == 0
    v.close()
    return ms / (8 * G) * 1e3


for rep in range(REPS):
    for staggered in (False, True):
        for capture in (False, True):
            us = run(staggered, capture)
            print('%-12s capture %-3s  %6.3f us per batched step (device)' % ('staggered' if staggered else 'synchronised', 'on' if capture else 'off', us), flush=True)
