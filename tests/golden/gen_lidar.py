#!/usr/bin/env python3
"""Golden vectors for the LidarInFront observation (reference gym_novel_gridworlds/observation_wrappers.py:10-80).

TEST INFRASTRUCTURE, same recipe as gen_golden.py (imports the unmodified reference through oracle/gym_shim):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=oracle/gym_shim:/root/reference python3 tests/golden/gen_lidar.py

For every configuration it re-uses the injected single-step states of <cfg>.npz (ss_pre_*): inject the state, call
step(action) on the LidarInFront-wrapped env and record the returned observation vector (lidar beams of the POST-step
state + inventory).  Also records the wrapper's static tables (lidar_items_id, max_beam_range, observation_space)."""
import json
import os
import sys

import numpy as np

import gym
import gym_novel_gridworlds  # noqa: F401
from gym_novel_gridworlds.novelty_wrappers import inject_novelty
from gym_novel_gridworlds.observation_wrappers import LidarInFront

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import CFGS, inject_state  # noqa: E402

LIDAR_CFGS = {'pogo10': 8, 'bow20': 8, 'axe10': 8, 'add32': 8, 'pogo13': 5, 'bowaxe16': 12, 'axeeasy10': 8}
N_CASES = {'add32': 300}


def main():
    out, meta = {}, {}
    for cfg, beams in LIDAR_CFGS.items():
        env_id, S, nov = CFGS[cfg]
        env = gym.make(env_id)
        env.map_size = S
        env = LidarInFront(env, num_beams=beams)        # observation wrapper first, novelty on top (tests/random_action.py:24-42)
        if nov is not None:
            env = inject_novelty(env, *nov)
        base = env.unwrapped
        np.random.seed(3)
        env.reset()
        lidar = env
        while not isinstance(lidar, LidarInFront):
            lidar = lidar.env
        g = np.load(os.path.join(HERE, cfg + '.npz'))
        n = min(N_CASES.get(cfg, 2000), len(g['ss_action']))
        obs = []
        for c in range(n):
            inject_state(base, g['ss_pre_map'][c], g['ss_pre_loc'][c], g['ss_pre_facing'][c], g['ss_pre_sel'][c], g['ss_pre_inv'][c])
            o, r, d, info = env.step(int(g['ss_action'][c]))
            assert r == g['ss_reward'][c] and d == bool(g['ss_done'][c])
            obs.append(np.asarray(o, np.int32))
        out[cfg + '_obs'] = np.array(obs, np.int32)
        meta[cfg] = {'num_beams': beams, 'n_cases': n, 'lidar_items_id': {k: int(v) for k, v in lidar.lidar_items_id.items()},
                     'max_beam_range': int(lidar.max_beam_range), 'obs_len': int(len(obs[0])),
                     'space_shape': list(lidar.observation_space.shape),
                     'inventory_order': [i for i in sorted(base.inventory_items_quantity) if i not in base.unbreakable_items]}
        print(cfg, meta[cfg]['obs_len'], meta[cfg]['max_beam_range'], flush=True)
    np.savez_compressed(os.path.join(HERE, 'lidar.npz'), **out)
    json.dump(meta, open(os.path.join(HERE, 'lidar.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
