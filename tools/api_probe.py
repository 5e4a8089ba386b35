"""Where a host-API step of a big batch spends its time: the whole VecNovelGridworld.step(), the C-ABI call alone with and
without the decoded info arrays, the host-side action copy."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, reset_prefetch=0)
v.reset()
acts = np.random.randint(0, 17, size=(64, n)).astype(np.int32)
for i in range(10):
    v.step(acts[i])
K = 100


def timed(f):
    t = time.perf_counter()
    for i in range(K):
        f(i)
    return (time.perf_counter() - t) / K * 1e6


print('step()                      %.1f us' % timed(lambda i: v.step(acts[i % 64])))
args = list(v._step_args[True])
L = _cabi.lib()
print('C call, info decoded        %.1f us' % timed(lambda i: L.ngw_step_host(*args)))
noinfo = list(args)
for k in (8, 9, 10, 11):
    noinfo[k] = None
print('C call, no info arrays      %.1f us' % timed(lambda i: L.ngw_step_host(*noinfo)))
print('pinned action copy          %.1f us' % timed(lambda i: v._act_pinned.__setitem__(Ellipsis, acts[i % 64])))
