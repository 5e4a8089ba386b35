"""Where the single-env adapter's step() goes: each stage of envs.py:step timed alone over the same action sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gym_novel_gridworlds_amd as G

env = G.make('NovelGridworld-Pogostick-v1')
env.reset()
rs = np.random.RandomState(0)
acts = [int(a) for a in rs.randint(0, 17, size=3000)]
for a in acts[:300]:
    env.step(a)
N = len(acts)


def timed(label, fn):
    t = time.perf_counter()
    for a in acts:
        fn(a)
    dt = (time.perf_counter() - t) / N * 1e6
    print('%-46s %6.2f us' % (label, dt), flush=True)
    return dt


timed('env.step()', env.step)
vec = env._backend()
timed('  _tables() == snapshot', lambda a: env._tables() == env._vec_fp)
timed('  _backend()', lambda a: env._backend())
timed('  _push() (nothing changed)', lambda a: env._push(vec))
timed('  vec.step1() (C-ABI ngw_step_host)', vec.step1)
fn, args = vec._step1_fn, vec._step1_args
timed('    the ctypes call alone', lambda a: fn(*args))
timed('  _pull()', lambda a: env._pull(vec))
timed('  get_observation + update_block_in_front', lambda a: (env.get_observation(), env.update_block_in_front()))
timed('  format_message', lambda a: env._spec.format_message(a, 0, 0))
t = time.perf_counter()
for i in range(300):
    env.reset()
print('env.reset() %.1f us' % ((time.perf_counter() - t) / 300 * 1e6))
