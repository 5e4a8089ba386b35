"""Fused rollout driven by the caller's action rows (ngw_rollout_actions) beside the in-kernel random policy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_novel_gridworlds_amd import VecNovelGridworld
n, T = 65536, 1000
v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, reset_prefetch=0)
v.reset()
acts = torch.randint(0, 17, (T, n), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
v.rollout_actions(acts.data_ptr(), n, 100); v.sync()
v.timing_begin(); t = time.perf_counter(); v.rollout_actions(acts.data_ptr(), n, T); ms = v.timing_end(); dt = time.perf_counter() - t
print('rollout_actions (caller\'s [T, N] int32 rows in HBM): %.3f us per batched step -> %.2f G env-steps/s' % (ms / T * 1e3, n * T / dt / 1e9))
v.rollout(100, 1, 0); v.sync()
v.timing_begin(); t = time.perf_counter(); v.rollout(T, 1, 100); ms = v.timing_end(); dt = time.perf_counter() - t
print('rollout (in-kernel uniform policy):                    %.3f us per batched step -> %.2f G env-steps/s' % (ms / T * 1e3, n * T / dt / 1e9))
