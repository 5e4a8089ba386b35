"""Where the reference's random_action.py loop shape spends its time on the adapter: steps, reset with / without a map size change, the first step after a reset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gym_novel_gridworlds_amd as G
env = G.make('NovelGridworld-Pogostick-v1'); env.reset()
rs = np.random.RandomState(0); A = env.action_space.n
for _ in range(300): env.step(int(rs.randint(A)))
def t(f, n):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print('step                         %.1f us' % t(lambda: env.step(3), 2000))
print('reset (same size)            %.1f us' % t(lambda: env.reset(), 300))
def rs_step(): env.reset(); env.step(3)
print('reset + first step           %.1f us' % t(rs_step, 300))
sizes = [int(x) for x in rs.randint(10, 20, 400)]
it = iter(sizes)
def resize(): env.map_size = next(it); env.reset()
for s in range(10, 20): env.map_size = s; env.reset()     # every handle exists
print('size change + reset          %.1f us' % t(resize, 300))
it = iter(sizes)
def resize_step(): env.map_size = next(it); env.reset(); env.step(3)
print('size change + reset + step   %.1f us' % t(resize_step, 300))
env.close()
