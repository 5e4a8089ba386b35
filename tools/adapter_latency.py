"""Single-env gym.Env adapter: step() latency (host attributes pushed, one env stepped on the GPU, state pulled back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gym_novel_gridworlds_amd as G
env = G.make('NovelGridworld-Pogostick-v1')
env.reset()
rs = np.random.RandomState(0)
acts = rs.randint(0, 17, size=2000)
for a in acts[:200]: env.step(int(a))
t = time.perf_counter()
for a in acts: env.step(int(a))
dt = time.perf_counter() - t
print('adapter step() %.1f us -> %.0f env-steps/s' % (dt / len(acts) * 1e6, len(acts) / dt))
t = time.perf_counter()
for i in range(200): env.reset()
print('adapter reset() %.1f us' % ((time.perf_counter() - t) / 200 * 1e6))
vec = env._backend()
t = time.perf_counter()
for a in acts: vec.step1(int(a))
print('  of which the C-ABI call (ngw_step_host through ctypes) %.1f us' % ((time.perf_counter() - t) / len(acts) * 1e6))
