#!/usr/bin/env python3
"""Drop-in differential against the LIVE reference (TEST INFRASTRUCTURE; build container only - the reference never travels).

    PYTHONPATH=oracle/gym_shim:/root/reference:.:tests python3 tests/dropin_driver.py wrappers|random [seed] [cases]

Run by tests/test_reference_dropin.py in a subprocess (the reference needs the stand-in `gym` package on the path, and with it
importable this package's adapter derives from that `gym.Env`, as it would under a real gym).  Two checks:

  wrappers  the reference's OWN wrapper classes - `inject_novelty` (novelty_wrappers.py:1586-1674), `LimitActions` (wrappers.py:57-85),
            `LidarInFront` / `AgentMap` (observation_wrappers.py:10-129) - stacked UNCHANGED on this package's single-env adapter, against
            the same stack on the reference's env: observation, reward, done, info dict, map and inventory after every step.
  random    randomised configurations nobody captured a fixture for - env id x map size x one or two novelties drawn per case - this
            package's own `inject_novelty` on the adapter against the reference's on its env; where the reference refuses the arguments
            (AssertionError) this package must refuse them with the same text.

Both drive the adapter on the CPU oracle backend (tests/ngw_testlib.OracleVec): what is checked is the drop-in boundary - the attribute
surface the wrappers poke at, the adapter's host logic, the spec compiler - and the oracle itself on configurations without fixtures; the
HIP kernels are held to the same oracle by the -m gpu tests.  States are injected reference -> adapter after every reset (the two sides
draw their maps from different generators)."""
import os
import sys

os.environ.setdefault('MPLBACKEND', 'Agg')
import numpy as np

import gym                                                     # the stand-in (oracle/gym_shim), or a real gym 0.18
import gym_novel_gridworlds                                    # noqa: F401  (registers the reference's ids)
from gym_novel_gridworlds import novelty_wrappers as RN
from gym_novel_gridworlds import observation_wrappers as RO
from gym_novel_gridworlds import wrappers as RW

import gym_novel_gridworlds_amd as G
import ngw_testlib as T

POGO, BOW, POGO0, BOW0 = 'NovelGridworld-Pogostick-v1', 'NovelGridworld-Bow-v1', 'NovelGridworld-Pogostick-v0', 'NovelGridworld-Bow-v0'


def unwrap(env):
    """Innermost env of a wrapper stack: follow `.env` (gym.Wrapper and this package's NoveltyWrapper both keep the wrapped env there;
    the base envs of both packages hold `env = None`, the curriculum argument of their constructors)."""
    seen = 0
    while getattr(env, 'env', None) is not None and seen < 16:
        env, seen = env.env, seen + 1
    return env


def make_pair(env_id, S):
    # (the reference's class itself, not gym.make: both packages register the same ids and the later registration wins)
    import gym_novel_gridworlds.envs as RE
    ref = {POGO: RE.PogostickV1Env, BOW: RE.BowV1Env, POGO0: RE.PogostickV0Env, BOW0: RE.BowV0Env}[env_id]()
    assert type(ref).__module__.startswith('gym_novel_gridworlds.envs')
    ours = G.make(env_id)
    ours._make_backend = lambda spec, seed_: T.OracleVec(spec, 1, seed=seed_)
    ours.seed(11)
    ref.map_size = S
    ours.map_size = S
    return ref, ours


def copy_state(ref, ours):
    """reference -> adapter, by direct attribute mutation like the reference's users (tests/keyboard_interface.py:93-100)."""
    rb, ob = unwrap(ref), unwrap(ours)
    assert rb.map.shape == ob.map.shape, (rb.map.shape, ob.map.shape)
    ob.map[...] = rb.map
    ob.agent_location = tuple(int(x) for x in rb.agent_location)
    ob.set_agent_facing(rb.agent_facing_str)
    ob.inventory_items_quantity = {k: int(v) for k, v in rb.inventory_items_quantity.items()}
    ob.selected_item = rb.selected_item
    ob.update_block_in_front()


def same_obs(a, b):
    if isinstance(a, dict):
        assert set(a) == set(b), (sorted(a), sorted(b))
        for k in a:
            if k == 'inventory_items_quantity':
                assert dict(a[k]) == dict(b[k]), (k, a[k], b[k])
            else:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (k, a[k], b[k])
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b)), (a, b)


def lockstep(ref, ours, steps, rs, tag):
    """Both stacks from the same state, the same actions; every observable compared after every step."""
    np.random.seed(int(rs.randint(1 << 30)))
    o1 = ref.reset()
    ours.reset()
    copy_state(ref, ours)
    rb, ob = unwrap(ref), unwrap(ours)
    nsteps = 0
    for t in range(steps):
        A = len(rb.actions_id)
        assert len(ob.actions_id) == A and dict(ob.actions_id) == dict(rb.actions_id), (tag, 'actions_id', rb.actions_id, ob.actions_id)
        a = int(rs.randint(A))
        if isinstance(ref, RW.LimitActions) and rs.randint(8):   # mostly ids the limited table knows (the rest must be refused alike)
            a = int(rs.randint(ref.action_space.n))
        if t % 37 == 5:                                        # random play never crafts: hand both sides the same items now and then
            names = [k for k in rb.inventory_items_quantity if k not in ('air', 'wall')]
            for k in rs.choice(names, size=min(3, len(names)), replace=False):
                q = int(rs.randint(0, 6))
                rb.inventory_items_quantity[k] += q
                ob.inventory_items_quantity[k] += q
        try:
            r1 = ref.step(a)
            err1 = None
        except (AssertionError, ValueError, KeyError, IndexError) as e:       # (e.g. LimitActions refusing an id)
            r1, err1 = None, (type(e).__name__, str(e))
        try:
            r2 = ours.step(a)
            err2 = None
        except (AssertionError, ValueError, KeyError, IndexError) as e:
            r2, err2 = None, (type(e).__name__, str(e))
        assert err1 == err2, (tag, t, a, err1, err2)
        if err1:
            continue
        nsteps += 1
        (ob1, rw1, d1, i1), (ob2, rw2, d2, i2) = r1, r2
        try:
            same_obs(ob1, ob2)
        except AssertionError as e:
            raise AssertionError((tag, t, a, 'observation', e.args)) from None
        assert rw1 == rw2 and bool(d1) == bool(d2), (tag, t, a, rw1, rw2, d1, d2)
        assert i1 == i2 and type(i1['step_cost']) is type(i2['step_cost']), (tag, t, a, i1, i2)
        assert np.array_equal(rb.map, ob.map), (tag, t, a, 'map')
        assert dict(rb.inventory_items_quantity) == dict(ob.inventory_items_quantity), (tag, t, a, rb.inventory_items_quantity, ob.inventory_items_quantity)
        assert tuple(rb.agent_location) == tuple(ob.agent_location) and rb.agent_facing_str == ob.agent_facing_str and rb.selected_item == ob.selected_item, (tag, t, a)
        if d1 or t % 97 == 96:
            np.random.seed(int(rs.randint(1 << 30)))
            ref.reset()
            ours.reset()
            copy_state(ref, ours)
    return nsteps


WRAPPER_CASES = [   # (env id, map size, observation wrapper or None, [(novelty, difficulty, arg1, arg2), ...], LimitActions set or None)
    (POGO, 10, 'lidar', [('axe', 'medium', 'wooden', '')], None),
    (POGO, 12, 'lidar', [('additem', 'hard', 'arrow', '')], None),
    (POGO, 10, 'lidar', [('firewall', 'hard', '', '')], None),
    (POGO, 11, None, [('crate', 'medium', '', '')], None),
    (POGO, 12, 'lidar', [('fencerestriction', 'hard', 'oak', '')], None),
    (POGO, 10, None, [('breakincrease', 'hard', '', '')], None),
    (POGO, 10, 'lidar', [('axe', 'hard', 'iron', 'true')], None),
    (POGO, 10, None, [('remapaction', 'hard', '', '')], None),
    (POGO, 10, 'lidar', [], None),
    (BOW, 12, 'lidar', [('extractincdec', 'hard', 'decrease', '')], None),
    (BOW, 11, 'agentmap', [], None),                      # (AgentMap under a wrapper that re-makes the observation: the reference itself raises TypeError)
    (BOW, 14, 'lidar', [('axetobreak', 'medium', 'wooden', '')], None),
    (POGO, 10, None, [('addchop', 'hard', '', '')], {'Forward', 'Left', 'Right', 'Break', 'Chop'}),
    (POGO, 13, 'agentmap', [('replaceitem', 'medium', 'wall', 'brick')], None),
    (POGO, 12, None, [('fence', 'hard', 'jungle', '')], {'Forward', 'Left', 'Right', 'Break', 'Craft_plank', 'Craft_stick'}),
]


def stack(env, obs_wrapper, novs, limit, seed):
    """The reference's own classes, applied in the reference scripts' order (tests/random_action.py:24-42: observation wrapper first,
    novelties on top); injection draws from the global numpy stream (Crate's contents, the remapped ids): same seed on both sides."""
    if obs_wrapper == 'lidar':
        env = RO.LidarInFront(env, num_beams=8)
    elif obs_wrapper == 'agentmap':
        env = RO.AgentMap(env)
    for i, nov in enumerate(novs):
        np.random.seed(seed + i)
        env = RN.inject_novelty(env, *nov)
    if limit is not None:
        env = RW.LimitActions(env, limit)
    return env


def run_wrappers(seed, steps):
    total = 0
    for ci, (env_id, S, ow, novs, limit) in enumerate(WRAPPER_CASES):
        ref, ours = make_pair(env_id, S)
        ref, ours = stack(ref, ow, novs, limit, seed + 100 * ci), stack(ours, ow, novs, limit, seed + 100 * ci)
        tag = '%s %dx%d %s %s %s' % (env_id, S, S, ow, novs, sorted(limit) if limit else None)
        n = lockstep(ref, ours, steps, np.random.RandomState(seed + ci), tag)
        total += n
        print('ok  %-110s %5d steps' % (tag, n), flush=True)
    print('WRAPPERS_OK %d cases %d steps' % (len(WRAPPER_CASES), total))


NOVELTY_POOL = [
    ('addchop', 'hard', '', ''), ('addjump', 'hard', '', ''), ('additem', 'easy', 'arrow', ''), ('additem', 'medium', 'gold', ''),
    ('additem', 'hard', 'paper', ''), ('additem', 'hard', '', ''), ('axe', 'easy', 'wooden', ''), ('axe', 'medium', 'iron', 'true'),
    ('axe', 'hard', 'wooden', 'false'), ('axe', 'medium', 'stone', ''), ('axe', 'easy', 'iron', 'maybe'), ('axetobreak', 'easy', 'iron', ''),
    ('axetobreak', 'medium', 'wooden', ''), ('axetobreak', 'hard', 'iron', ''), ('breakincrease', 'hard', '', ''),
    ('breakincrease', 'hard', 'tree_log', ''), ('crate', 'easy', '', ''), ('crate', 'hard', '', ''),
    ('extractincdec', 'hard', 'decrease', ''), ('extractincdec', 'hard', 'increase', ''), ('extractincdec', 'hard', 'sideways', ''),
    ('fence', 'easy', 'oak', ''), ('fence', 'hard', 'jungle', ''), ('fence', 'medium', '', ''), ('fencerestriction', 'medium', 'oak', ''),
    ('fencerestriction', 'hard', 'birch', ''), ('firewall', 'easy', '', ''), ('firewall', 'medium', '', ''), ('firewall', 'hard', '', ''),
    ('remapaction', 'easy', '', ''), ('remapaction', 'medium', '', ''), ('remapaction', 'hard', '', ''),
    ('replaceitem', 'easy', 'wall', 'brick'), ('replaceitem', 'medium', 'tree_log', 'oak_log'), ('replaceitem', 'hard', 'wall', ''),
    ('crate', 'extreme', '', ''), ('teleport', 'hard', '', ''),
]


INVALID = {('additem', 'hard', '', ''), ('axe', 'medium', 'stone', ''), ('axe', 'easy', 'iron', 'maybe'), ('extractincdec', 'hard', 'sideways', ''),
           ('fence', 'medium', '', ''), ('replaceitem', 'hard', 'wall', ''), ('crate', 'extreme', '', ''), ('teleport', 'hard', '', '')}


def inject(fn, env, nov, seed):
    np.random.seed(seed)
    try:
        return fn(env, *nov), None
    except AssertionError as e:
        return env, ('AssertionError', str(e))
    except AttributeError as e:                               # (breakincrease with an unknown item: the reference's assert message itself raises, :1634)
        return env, ('AttributeError', str(e))


def run_random(seed, cases, steps):
    rs = np.random.RandomState(seed)
    done, refused, stepped = 0, 0, 0
    while done < cases:
        env_id = [POGO, POGO, BOW, BOW, POGO0, BOW0][int(rs.randint(6))]
        S = int(rs.randint(9, 19))
        novs = [NOVELTY_POOL[int(rs.randint(len(NOVELTY_POOL)))] for _ in range(1 + int(rs.randint(2)))]
        if any(n in INVALID for n in novs) and rs.randint(3):      # (entries the reference refuses stay in, at a lower rate)
            continue
        if len(novs) == 2:
            names = {n[0] for n in novs}
            # stacks the reference itself cannot run, or runs with its own wrapper-shadowing quirks that DESIGN.md lists as not reproduced
            if len(names) == 1 or 'remapaction' in names or names & {'fence', 'fencerestriction'} and names & {'firewall', 'replaceitem'} or \
               sum(n[0] in ('axe', 'axetobreak', 'breakincrease', 'addchop') for n in novs) == 2:
                continue
        ref, ours = make_pair(env_id, S)
        tag = '%s %dx%d %s' % (env_id, S, S, novs)
        bad = False
        for i, nov in enumerate(novs):
            ref, e1 = inject(RN.inject_novelty, ref, nov, seed + 7 * done + i)
            ours, e2 = inject(G.inject_novelty, ours, nov, seed + 7 * done + i)
            assert e1 == e2, (tag, 'argument errors differ', e1, e2)
            bad = bad or e1 is not None
        done += 1
        if bad:
            refused += 1
            print('ok  %-100s refused alike' % tag, flush=True)
            continue
        try:
            n = lockstep(ref, ours, steps, np.random.RandomState(seed + done), tag)
        except AssertionError as e:
            if 'Cannot place items' in str(e):               # a map too small for this stack: both sides must say so
                print('ok  %-100s placement exhausted' % tag, flush=True)
                continue
            raise
        stepped += n
        print('ok  %-100s %5d steps' % (tag, n), flush=True)
    print('RANDOM_OK %d cases (%d refused alike) %d steps' % (done, refused, stepped))


if __name__ == '__main__':
    mode = sys.argv[1]
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    if mode == 'wrappers':
        run_wrappers(seed, int(sys.argv[3]) if len(sys.argv) > 3 else 600)
    else:
        run_random(seed, int(sys.argv[3]) if len(sys.argv) > 3 else 24, int(sys.argv[4]) if len(sys.argv) > 4 else 400)
