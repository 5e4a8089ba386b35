"""Duration of the ONE step launch in which every env's episode ends (synchronized horizons) and the whole batch takes its
prepared next episode - against a plain step launch.  HIP event pair per launch, median over 40 repetitions."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, apply_novelty, make_spec

for wl in (sys.argv[1:] or ['C2', 'C3', 'C5']):
    env_id, S, nov, n, desc = bench.WORKLOADS[wl]
    spec = make_spec(env_id, S)
    if nov:
        apply_novelty(spec, *nov)
    A = len(spec.actions_id)
    H = 100
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=1, autoreset=True, horizon=H, reset_prefetch=1 << 20)   # refills only after explicit resets
    acts = torch.randint(0, A, (4, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    plain, ending = [], []
    for rep in range(40):
        v.reset(); v.sync()                                   # reset + refill: every env has its next episode prepared
        v.step_device(acts[0].data_ptr()); v.sync()
        v.timing_begin(); v.step_device(acts[1].data_ptr()); plain.append(v.timing_end() * 1e3)
        v.set_state(0, step_count=np.full(n, H - 1, np.int32))
        v.timing_begin(); v.step_device(acts[2].data_ptr()); ending.append(v.timing_end() * 1e3)
        assert int(v.get_state(0, 1)['episode'][0]) == 2 * rep + 2
    print('%s: plain step launch %.2f us, the step in which all %d envs take their prepared episode %.2f us (median of 40)'
          % (wl, np.median(plain), n, np.median(ending)), flush=True)
    v.close()
