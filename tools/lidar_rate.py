"""Time the LidarInFront observation at C2 (65 536 envs, 10 x 10, 8 beams), device-resident: the plain step, the stand-alone lidar
launch, and the step with the FUSED epilogue in every row format (int32 / int16 / packed) and with both marches (world-frame
rays, per-lane table: NGW_LIDAR_WORLD=0), each as a 64-step hipGraph replayed 16 times (default prepared-episode cadence).
    python tools/lidar_rate.py [n_envs] [workload: C2 | C3 | C5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi, make_spec  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else 'C2'
if wl == 'C5':                                        # bench.py's C5: Pogostick-v1 + inject_novelty('additem','hard','arrow'), 32 x 32
    from gym_novel_gridworlds_amd import apply_novelty
    spec = make_spec('NovelGridworld-Pogostick-v1', 32)
    np.random.seed(0)
    apply_novelty(spec, 'additem', 'hard', 'arrow', '')
else:
    spec = make_spec('NovelGridworld-Pogostick-v1', 10) if wl == 'C2' else make_spec('NovelGridworld-Bow-v1', 20)
A = len(spec.actions_id)
L = _cabi.lib()
acts = None


def replay(v, label, reps=16, G=64):
    global acts
    if acts is None:
        acts = torch.randint(0, A, (G, n), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
    v.reset()
    v.graph_build(acts.data_ptr(), n, G)
    v.graph_launch(4)
    v.sync()
    v.timing_begin()
    t = time.perf_counter()
    v.graph_launch(reps)
    ms = v.timing_end()
    dt = time.perf_counter() - t
    assert v.error_flags() == 0
    print('%-58s %6.2f us per batched step (device %6.2f) -> %5.2f G env-steps/s' % (label, dt / (reps * G) * 1e6, ms / (reps * G) * 1e3, n * reps * G / dt / 1e9), flush=True)
    return ms / (reps * G) * 1e3


def env():
    return VecNovelGridworld(spec=spec, num_envs=n, autoreset=True, horizon=100)


# clock warm-up on a scratch handle
s = env(); s.reset()
t_end = time.perf_counter() + 0.25
while time.perf_counter() < t_end:
    s.rollout(200, 1, 0); s.sync()
s.close()

v = env()
replay(v, 'plain step')
v.close()
for world in (('', 'B0', '1', '0') if os.environ.get('NGW_LIDAR_SWEEP', '1') != '0' else ('',)):
    os.environ.pop('NGW_LIDAR_WORLD', None)
    os.environ.pop('NGW_LIDAR_BOARDS', None)
    if world == 'B0':                                 # A/B: the marches over maps staged through LDS instead of the occupancy bit rows
        os.environ['NGW_LIDAR_BOARDS'] = '0'
    elif world:
        os.environ['NGW_LIDAR_WORLD'] = world
    for dt_, name in ((np.int32, 'int32'), (np.int16, 'int16'), ('packed', 'packed u8+i16')):
        v = env()
        v.lidar_configure(num_beams=8, fused=True, dtype=dt_)
        replay(v, 'fused lidar, %s rows (%d B/env), %s rays' % (name, v.lidar_row_bytes, {'': 'default: occupancy bit rows, in-place step' if v.step_reads_map_in_place else 'default march', 'B0': 'staged march (constant-offset at 10x10)', '1': 'world-frame march', '0': 'per-lane table march'}[world]))
        if world == '':
            v.rollout(100, 1, 0); v.sync()
            v.timing_begin(); v.rollout(1000, 1, 100); ms = v.timing_end()
            print("    fused rollout (1000 steps), observation of the final state: %.3f us per batched step -> %.1f G env-steps/s" % (ms, n / ms / 1e3), flush=True)
        v.close()
os.environ.pop('NGW_LIDAR_WORLD', None)
os.environ.pop('NGW_LIDAR_BOARDS', None)
v = env()
v.lidar_configure(num_beams=8, fused=False, dtype=np.int16)
v.reset()
for k in range(20):
    L.ngw_lidar(v._h)
v.sync()
K = 500
v.timing_begin(); t = time.perf_counter()
for k in range(K):
    L.ngw_lidar(v._h)
ms = v.timing_end(); dt = time.perf_counter() - t
print('stand-alone lidar launch (int16 rows): %.2f us per launch (device %.2f)' % (dt / K * 1e6, ms / K * 1e3))
v.close()
