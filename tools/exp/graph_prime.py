"""Is the first-replay penalty of an instantiated graph per exec or per stream?  Build the K-step graph, optionally replay another (empty-kernel) graph on the same stream first, time the first regions."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, make_spec, _cabi
K = 20
env_id, S, nov, n, desc = bench.WORKLOADS['C2']
spec = make_spec(env_id, S)
A = len(spec.actions_id)
fl = _cabi.lib().ngw_debug_launch_floor
fl.argtypes, fl.restype = [C.c_void_p, C.c_int32, C.c_int, C.POINTER(C.c_double)], C.c_int
for prime in (0, 1, 0, 1):
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
    v.reset()
    acts = torch.randint(0, A, (K, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.step_device_many(acts[0].data_ptr(), n, 5); v.sync()
    v.graph_build(acts[0].data_ptr(), n, K)
    if prime:
        us = C.c_double(0)
        _cabi.check(fl(v._h, K, 1, C.byref(us)))
    ts = []
    for rep in range(6):
        v.sync(); t0 = time.perf_counter(); v.graph_launch(1); v.sync(); ts.append((time.perf_counter() - t0) * 1e6)
    print('prime %d: regions %s us' % (prime, ' '.join('%.1f' % x for x in ts)), flush=True)
    v.close()
