"""Step rate when episode ends are STAGGERED (the training regime: ~1 % of the envs reset in every batched step, spread
over all waves) beside the synchronized-horizon regime bench.py measures.  Usage: stagger_rate.py [n_envs] [horizon]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_novel_gridworlds_amd import VecNovelGridworld  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prefetch = int(os.environ.get('NGW_PREFETCH', '0'))
acts = torch.randint(0, 17, (64, n), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
for name, stagger in (('synchronized', False), ('staggered', True)):
    v = VecNovelGridworld(num_envs=n, autoreset=True, horizon=H, seed=3, reset_prefetch=0)
    if prefetch and hasattr(v, 'set_reset_prefetch'):
        v.set_reset_prefetch(prefetch)
    v.reset()
    if stagger:
        v.set_state(0, step_count=(np.arange(n) * 7919 % H).astype(np.int32))
    v.graph_build(acts.data_ptr(), n, 64)
    v.graph_launch(4); v.sync()
    reps = 16
    v.timing_begin(); t = time.perf_counter(); v.graph_launch(reps); ms = v.timing_end(); dt = time.perf_counter() - t
    steps = reps * 64
    print('%s horizon %d%s: %.2f us per batched step (device %.2f) -> %.2f G env-steps/s'
          % (name, H, ' prefetch every %d' % prefetch if prefetch else '', dt / steps * 1e6, ms / steps * 1e3, n * steps / dt / 1e9), flush=True)
    v.close()
