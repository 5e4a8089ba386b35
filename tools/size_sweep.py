"""Per-launch step period of Pogostick-v1 over map sizes (65 536 envs, synchronized horizon 100, prepared episodes on, graph replay):
no cliff between the staged kernel's three LDS layouts (S*S % 16 == 0 / % 4 == 0 / odd) and the no-stage kernel from 16x16 up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_novel_gridworlds_amd import VecNovelGridworld, make_spec

n = 65536
for S in [int(x) for x in (sys.argv[1:] or [10, 11, 12, 13, 14, 15, 16, 20, 24, 32, 40, 48, 64])]:
    spec = make_spec('NovelGridworld-Pogostick-v1', S)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
    v.reset()
    if os.environ.get('NGW_STAGGER') == '1':                 # episode ends spread over the batch
        v.set_state(0, step_count=(np.arange(n) * 7919 % 100).astype(np.int32))
    acts = torch.randint(0, A, (100, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.graph_build(acts.data_ptr(), n, 100)
    v.graph_launch(3); v.sync()
    v.timing_begin(); v.graph_launch(10); ms = v.timing_end()
    print('%2dx%-2d  %.2f us per batched step  %.2f G env-steps/s' % (S, S, ms, n / ms / 1e3), flush=True)   # (1000 steps: ms per 1000 = us per step)
    v.close()
