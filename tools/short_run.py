"""Wall-clock of a SHORT timed region (the driver runs bench.py --steps 20 --warmup 5): K one-step launches issued eagerly from
one C-ABI call vs one replay of a K-node hipGraph, fence to fence (stream sync on both sides), median of 30 regions."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, make_spec

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
env_id, S, nov, n, desc = bench.WORKLOADS['C2']
spec = make_spec(env_id, S)
A = len(spec.actions_id)
v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
v.reset()
acts = torch.randint(0, A, (K, n), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
for name in ('eager', 'graph'):
    if name == 'graph':
        v.graph_build(acts[0].data_ptr(), n, K)
    ts, enq = [], []
    for rep in range(40):
        v.sync()
        t0 = time.perf_counter()
        if name == 'eager':
            v.step_device_many(acts[0].data_ptr(), n, K)
        else:
            v.graph_launch(1)
        t1 = time.perf_counter()
        v.sync()
        t2 = time.perf_counter()
        ts.append(t2 - t0); enq.append(t1 - t0)
    first = ts[0] * 1e6
    ts, enq = np.array(ts[10:]) * 1e6, np.array(enq[10:]) * 1e6
    print('%s K=%d: region %.1f us median (%.2f us/step), min %.1f; enqueue call %.1f us; FIRST region %.1f us' % (name, K, np.median(ts), np.median(ts) / K, ts.min(), np.median(enq), first), flush=True)
v.close()
