"""Experiment: the same 65 536 envs as 1 / 2 / 4 / 8 independent handles (own streams, own hipGraphs) stepped concurrently."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_novel_gridworlds_amd import VecNovelGridworld  # noqa: E402

N, K = 65536, 400
for split in (1, 2, 4, 8):
    n = N // split
    hs = [VecNovelGridworld(num_envs=n, autoreset=True, horizon=100, env_index_base=i * n, reset_prefetch=0) for i in range(split)]
    acts = [torch.randint(0, 17, (K, n), dtype=torch.int32, device='cuda') for _ in range(split)]
    torch.cuda.synchronize()
    for h, a in zip(hs, acts):
        h.reset()
        h.graph_build(a.data_ptr(), n, K)
        h.graph_launch(1)
    for h in hs:
        h.sync()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for h in hs:
            h.graph_launch(1)
        for h in hs:
            h.sync()
        best = min(best, time.perf_counter() - t)
    print('split %d x %d envs: %.2f us per batched step of %d envs -> %.2f G env-steps/s' % (split, n, best / K * 1e6, N, N * K / best / 1e9), flush=True)
    for h in hs:
        h.close()
