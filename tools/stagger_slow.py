"""How many resets of the staggered regime miss their prepared episode (diagnostics counter NgwNx::slow)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, make_spec
L = _cabi.lib()
L.ngw_debug_slow_resets.argtypes = [C.c_void_p]; L.ngw_debug_slow_resets.restype = C.c_longlong
env_id, S, nov, n, desc = bench.WORKLOADS['C2']
spec = make_spec(env_id, S)
A = len(spec.actions_id)
v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
for mode in ('explicit 32', 'default'):
    if mode.startswith('explicit'):
        v.set_reset_prefetch(32)
    v.reset()
    v.set_state(0, step_count=(np.arange(n) * 7919 % 100).astype(np.int32))
    acts = torch.randint(0, A, (64, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    s0 = L.ngw_debug_slow_resets(v._h)
    v.graph_build(acts.data_ptr(), n, 64)
    for rep in range(3):
        v.timing_begin(); v.graph_launch(8); ms = v.timing_end()
        s1 = L.ngw_debug_slow_resets(v._h)
        print('%s: graph 8 x 64 steps: %.2f us per step, slow resets so far %d (of ~%d resets)' % (mode, ms * 1e3 / 512, s1 - s0, 512 * (rep + 1) * n // 100), flush=True)
    v.timing_begin()
    for i in range(256):
        v.step_device(acts[i % 64].data_ptr())
    ms = v.timing_end()
    print('%s: eager 256 steps: %.2f us per step, slow resets so far %d' % (mode, ms * 1e3 / 256, L.ngw_debug_slow_resets(v._h) - s0), flush=True)
v.close()
