"""Explicit resets of every env of a workload, prepared episodes off (every launch BUILDS its episodes): the command the
new-episode kernel's HBM counters are collected on (tools/profile_round.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_novel_gridworlds_amd import VecNovelGridworld, apply_novelty, make_spec

wl = sys.argv[1] if len(sys.argv) > 1 else 'C5'
env_id, S, nov, n, desc = bench.WORKLOADS[wl]
spec = make_spec(env_id, S)
if nov:
    apply_novelty(spec, *nov)
v = VecNovelGridworld(spec=spec, num_envs=n, seed=1, reset_prefetch=0)
for _ in range(6):
    v.reset()
v.sync()
assert v.error_flags() == 0
print('%s: 6 resets of %d envs' % (wl, n))
