"""Which refill cadence a graph captures after a 100-step eager warm-up (diagnostics)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, apply_novelty, make_spec
L = _cabi.lib()
L.ngw_debug_slow_resets.argtypes = [C.c_void_p]; L.ngw_debug_slow_resets.restype = C.c_longlong
L.ngw_debug_refill_cadence.argtypes = [C.c_void_p]; L.ngw_debug_refill_cadence.restype = C.c_int
for wl in sys.argv[1:] or ['X1', 'X3', 'C2']:
    env_id, S, nov, n, desc = bench.WORKLOADS[wl]
    spec = make_spec(env_id, S)
    if nov:
        apply_novelty(spec, *nov)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
    v.reset()
    acts = torch.randint(0, A, (200, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.step_device_many(acts.data_ptr(), n, 100)
    v.sync()
    print(wl, 'after 100 eager steps: cadence', L.ngw_debug_refill_cadence(v._h), 'misses', L.ngw_debug_slow_resets(v._h))
    v.graph_build(acts[100].data_ptr(), n, 100)
    print(wl, 'captured with cadence', L.ngw_debug_refill_cadence(v._h))
    v.close()
