#!/usr/bin/env python3
"""Golden vectors for the AgentMap observation (reference gym_novel_gridworlds/observation_wrappers.py:83-129).

TEST INFRASTRUCTURE, same recipe as gen_golden.py / gen_lidar.py (imports the unmodified reference through
oracle/gym_shim):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=oracle/gym_shim:/root/reference python3 tests/golden/gen_agentmap.py

For every configuration it re-uses the injected single-step states of <cfg>.npz (ss_pre_*): inject the state, call
step(action) on the AgentMap-wrapped env and record the returned window, facing id and inventory."""
import os
import sys

import numpy as np

import gym
import gym_novel_gridworlds  # noqa: F401
from gym_novel_gridworlds.novelty_wrappers import inject_novelty
from gym_novel_gridworlds.observation_wrappers import AgentMap

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import CFGS, inject_state  # noqa: E402

AGENTMAP_CFGS = ['pogo10', 'bow20', 'add32', 'pogo13']      # axe + AgentMap raises TypeError in the reference (AxeMedium.step calls observation() without obs)
N_CASES = 400


def main():
    out = {}
    for cfg in AGENTMAP_CFGS:
        env_id, S, nov = CFGS[cfg]
        env = gym.make(env_id)
        env.map_size = S
        env = AgentMap(env)
        if nov is not None:
            env = inject_novelty(env, *nov)
        base = env.unwrapped
        np.random.seed(3)
        env.reset()
        g = np.load(os.path.join(HERE, cfg + '.npz'))
        n = min(N_CASES, len(g['ss_action']))
        views, facing, inv = [], [], []
        for c in range(n):
            inject_state(base, g['ss_pre_map'][c], g['ss_pre_loc'][c], g['ss_pre_facing'][c], g['ss_pre_sel'][c], g['ss_pre_inv'][c])
            o, r, d, info = env.step(int(g['ss_action'][c]))
            assert r == g['ss_reward'][c] and d == bool(g['ss_done'][c])
            assert o['agent_map'].shape == (11, 11)
            views.append(np.asarray(o['agent_map'], np.int8))
            facing.append(int(o['agent_facing_id']))
            inv.append([int(o['inventory_items_quantity'][k]) for k in sorted(base.items_id, key=base.items_id.get)])
        out[cfg + '_view'] = np.array(views, np.int8)
        out[cfg + '_facing'] = np.array(facing, np.int32)
        out[cfg + '_inv'] = np.array(inv, np.int32)
        out[cfg + '_dtype'] = np.array(str(o['agent_map'].dtype))
        sp = None
        w = env
        while not isinstance(w, AgentMap):
            w = w.env
        out[cfg + '_space_shape'] = np.array(w.observation_space.spaces['agent_map'].shape)
        print(cfg, n, o['agent_map'].dtype, flush=True)
    np.savez_compressed(os.path.join(HERE, 'agentmap.npz'), **out)


if __name__ == '__main__':
    main()
