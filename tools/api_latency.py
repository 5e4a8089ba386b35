"""Host-API step latency (actions in, observation + outputs out, all through PCIe) by batch size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_novel_gridworlds_amd import VecNovelGridworld
for n in [int(x) for x in os.environ.get('NGW_API_SIZES', '1,64,1024,16384,65536').split(',')]:
    v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=100)
    v.reset()
    a = np.random.randint(0, 17, size=(64, n)).astype(np.int32)
    for k in range(50): v.step(a[k % 64])
    K = 300 if n <= 16384 else 60
    t = time.perf_counter()
    for k in range(K): v.step(a[k % 64])
    dt = time.perf_counter() - t
    print('n = %6d  step() %.1f us  -> %.2f M env-steps/s' % (n, dt / K * 1e6, n * K / dt / 1e6), flush=True)
    v.close()
