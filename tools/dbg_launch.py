"""Diagnostic: launch floor (empty kernel) vs staging only vs full step, wall and HIP-event time per launch."""
import sys, time, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi
import torch
L = _cabi.lib()
L.ngw_debug_launch.argtypes = [C.c_void_p, C.c_int, C.c_int32]
if sys.argv[1:] == ['calib']:
    # PMC calibration: 20 staging-only launches at 1 Mi envs (state 2 x 148 MB, beyond the 256 MiB Infinity Cache
    # together with the output buffers); bytes per launch are known: read = write = n_pad * (S*S + 4*K + 12) (+ small)
    n = 1 << 20
    v = VecNovelGridworld(num_envs=n)
    v.reset(); v.sync()
    L.ngw_debug_launch(v._h, 9, 20); v.sync()
    print('calib n', n, 'bytes_read_per_launch', n * 157, 'bytes_written_per_launch', n * 166)
    sys.exit(0)
for n in [int(x) for x in (sys.argv[1:] or ['65536', '262144', '1048576'])]:
    v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, reset_prefetch=0)
    v.reset()
    acts = torch.randint(0, 17, (64, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    for mode, name in ((8, 'nop'), (10, 'nop 256-thread WGs'), (11, 'nop 1024-thread WGs'), (12, 'nop 128-thread WGs + LDS'), (9, 'copy'), (0, 'step')):
        for rep in range(2):
            v.sync(); t = time.perf_counter()
            K = 500
            if mode == 0:
                for i in range(K): v.step_device(acts[i % 64].data_ptr())
            else:
                L.ngw_debug_launch(v._h, mode, K)
            v.sync(); dt = time.perf_counter() - t
        v.timing_begin()
        if mode == 0:
            for i in range(K): v.step_device(acts[i % 64].data_ptr())
        else:
            L.ngw_debug_launch(v._h, mode, K)
        ms = v.timing_end()
        print(n, name, 'wall us/launch %.2f' % (dt / K * 1e6), 'event us/launch %.2f' % (ms / K * 1e3), flush=True)
    v.graph_build(acts.data_ptr(), n, 64)
    v.graph_launch(2); v.sync()
    v.timing_begin(); t = time.perf_counter(); v.graph_launch(10); ms = v.timing_end(); dt = time.perf_counter() - t
    print(n, 'graph step', 'wall us/launch %.2f' % (dt / 640 * 1e6), 'event us/launch %.2f' % (ms / 640 * 1e3), flush=True)
    v.close()
