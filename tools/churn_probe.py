"""High-churn configuration (FireWall hard: ~1.3 % of the envs die per step): per-step time of the eager loop, the refill cadence
the handle adapts to and how many resets still miss their prepared episode."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, apply_novelty, make_spec
L = _cabi.lib()
L.ngw_debug_slow_resets.argtypes = [C.c_void_p]; L.ngw_debug_slow_resets.restype = C.c_longlong
L.ngw_debug_refill_cadence.argtypes = [C.c_void_p]; L.ngw_debug_refill_cadence.restype = C.c_int
wl = sys.argv[1] if len(sys.argv) > 1 else 'X1'
env_id, S, nov, n, desc = bench.WORKLOADS[wl]
spec = make_spec(env_id, S)
if nov:
    apply_novelty(spec, *nov)
A = len(spec.actions_id)
v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
v.reset()
acts = torch.randint(0, A, (64, n), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
ep0 = int(v.get_state()['episode'].sum())
for rep in range(12):
    s0 = L.ngw_debug_slow_resets(v._h)
    v.timing_begin()
    for i in range(256):
        v.step_device(acts[i % 64].data_ptr())
    ms = v.timing_end()
    ep1 = int(v.get_state()['episode'].sum())
    print('%s eager 256 steps: %.2f us per step; cadence now %d, depth %d; resets %d (%.2f %% of the envs per step), of which %d missed their prepared episode'
          % (wl, ms * 1e3 / 256, L.ngw_debug_refill_cadence(v._h), v.reset_prefetch_depth, ep1 - ep0, 100.0 * (ep1 - ep0) / 256 / n, L.ngw_debug_slow_resets(v._h) - s0), flush=True)
    ep0 = ep1
v.close()
