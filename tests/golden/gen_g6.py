#!/usr/bin/env python3
"""G6 distribution fixtures: what the reference's reset() of the shuffled-subset novelties looks like
over many episodes - per-cell frequencies of the pass item, the histogram of how many cells got it, per-cell frequencies
of the agent cell.  They pin the DISTRIBUTION of the device's reset passes, which draw the same uniformly random subset
without numpy's shuffle (include/ngw.h, ngw_spec.n_passes), against the reference itself.

TEST INFRASTRUCTURE, same recipe as gen_golden.py (imports the unmodified reference through oracle/gym_shim):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=oracle/gym_shim:/root/reference python3 tests/golden/gen_g6.py

Everything written is DATA (integer counts)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import CFGS, REMAP_SEED, make_env, novelty_list  # noqa: E402

# cfg -> (number of reference resets, name of the item the pass writes)
G6 = {'add32': (10000, 'arrow'), 'add12m': (10000, 'spring'), 'add11e': (10000, 'arrow'), 'crate12h': (10000, 'crate'),
      'fire14m': (10000, 'fire_wall'), 'fire10h': (4000, 'fire_wall'), 'replwall12e': (10000, 'brick')}
SEED = 20261004


def main():
    check = '--check' in sys.argv[1:]
    only = [a for a in sys.argv[1:] if a != '--check']
    bad = []
    for cfg, (n, item) in G6.items():
        if only and cfg not in only:
            continue
        env = make_env(cfg)
        base = env.unwrapped
        S = CFGS[cfg][1]
        np.random.seed(SEED)
        freq = np.zeros(S * S, np.int64)
        agent = np.zeros(S * S, np.int64)
        hist = np.zeros(S * S + 1, np.int64)
        item_id = None
        for _ in range(n):
            env.reset()
            if item_id is None:
                item_id = int(base.items_id[item])
            m = np.asarray(base.map).ravel() == item_id
            freq += m
            hist[int(m.sum())] += 1
            r, c = base.agent_location
            agent[int(r) * S + int(c)] += 1
        new = dict(n=np.int64(n), item=np.int64(item_id), freq=freq.astype(np.int32), hist=hist.astype(np.int32), agent=agent.astype(np.int32),
                   seed=np.int64(SEED))
        path = os.path.join(HERE, 'g6_%s.npz' % cfg)
        if check:                                        # compare with the committed file instead of writing it
            old = dict(np.load(path))
            bad += ['g6_%s: %s differs' % (cfg, k) for k in new if k not in old or not np.array_equal(new[k], old[k])]
        else:
            np.savez_compressed(path, **new)
        print(cfg, 'resets', n, 'item', item, item_id, 'mean cells', float(freq.sum()) / n)
    if check:
        if bad:
            sys.exit('FIXTURE CHECK FAILED:\n  ' + '\n  '.join(bad))
        print('fixture check ok: G6')


if __name__ == '__main__':
    main()
