"""Time the LidarInFront observation kernel (device-resident, one launch per batched step) beside the step kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, reset_prefetch=0)
v.reset()
v.lidar_configure(num_beams=8)
acts = torch.randint(0, 17, (64, n), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
L = _cabi.lib()
for k in range(20):
    v.step_device(acts[k].data_ptr()); L.ngw_lidar(v._h)
v.sync()
K = 500
v.timing_begin(); t = time.perf_counter()
for k in range(K):
    L.ngw_lidar(v._h)
ms = v.timing_end(); dt = time.perf_counter() - t
print('lidar only: %.2f us/launch (device %.2f), %d x %d int32 out = %.0f B/env' % (dt / K * 1e6, ms / K * 1e3, n, v.lidar_len, v.lidar_len * 4))
v.timing_begin(); t = time.perf_counter()
for k in range(K):
    v.step_device(acts[k % 64].data_ptr()); L.ngw_lidar(v._h)
ms = v.timing_end(); dt = time.perf_counter() - t
print('step + lidar: %.2f us per batched step (device %.2f) -> %.2f G env-steps/s' % (dt / K * 1e6, ms / K * 1e3, n * K / dt / 1e9))
v.lidar_configure(num_beams=8, fused=True)
for k in range(20):
    v.step_device(acts[k].data_ptr())
v.sync()
v.timing_begin(); t = time.perf_counter()
for k in range(K):
    v.step_device(acts[k % 64].data_ptr())
ms = v.timing_end(); dt = time.perf_counter() - t
print('step with FUSED lidar epilogue: %.2f us per batched step (device %.2f) -> %.2f G env-steps/s' % (dt / K * 1e6, ms / K * 1e3, n * K / dt / 1e9))
v.graph_build(acts.data_ptr(), n, 64); v.graph_launch(1); v.sync()
v.timing_begin(); t = time.perf_counter(); v.graph_launch(8); ms = v.timing_end(); dt = time.perf_counter() - t
print('  same, hipGraph replay: %.2f us per batched step -> %.2f G env-steps/s' % (dt / 512 * 1e6, n * 512 / dt / 1e9))
v.rollout(50, 1, 0); v.sync()
v.timing_begin(); t = time.perf_counter(); v.rollout(500, 1, 50); ms = v.timing_end(); dt = time.perf_counter() - t
print('fused rollout with lidar every step: %.2f us per batched step -> %.2f G env-steps/s' % (dt / 500 * 1e6, n * 500 / dt / 1e9))
import numpy as np
v.lidar_configure(num_beams=8, fused=True, dtype=np.int16)
for k in range(20):
    v.step_device(acts[k].data_ptr())
v.sync()
v.graph_build(acts.data_ptr(), n, 64); v.graph_launch(1); v.sync()
v.timing_begin(); t = time.perf_counter(); v.graph_launch(8); ms = v.timing_end(); dt = time.perf_counter() - t
print('int16 observation rows, fused, hipGraph replay: %.2f us per batched step -> %.2f G env-steps/s' % (dt / 512 * 1e6, n * 512 / dt / 1e9))
v.rollout(50, 1, 0); v.sync()
v.timing_begin(); t = time.perf_counter(); v.rollout(500, 1, 50); ms = v.timing_end(); dt = time.perf_counter() - t
print('int16, fused rollout with lidar every step: %.2f us per batched step -> %.2f G env-steps/s' % (dt / 500 * 1e6, n * 500 / dt / 1e9))
