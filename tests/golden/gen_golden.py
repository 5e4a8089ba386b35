#!/usr/bin/env python3
"""Golden-vector generator: imports the UNMODIFIED reference from /root/reference and
records inputs / expected outputs of its reset()/step() hot path as small fixtures.

TEST INFRASTRUCTURE.  Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
    PYTHONPATH=oracle/gym_shim:/root/reference python3 tests/golden/gen_golden.py

`oracle/gym_shim` is a stand-in for the reference's one missing third-party dependency
(`gym`), see its docstring.  Everything written is DATA (npz / json): states, actions,
rewards, flags, costs, message strings.  No reference source text is stored.

Fixture sets:
  G1  spec.json            ids / action tables / recipes / exceptions for every configuration
  G2  <cfg>.npz  rs_*      seeded resets (3 consecutive per seed) + RNG stream position after
  G3  <cfg>.npz  tr<k>_*   lock-step random traces with inventory injections (coverage recipe)
  G4  <cfg>.npz  ss_*      random single-step cases from injected states (exhaustive branches)
  G5  <cfg>.npz  so<k>_*   scripted-solver episodes that reach `done`
  C1  c1loop.npz           the tests/random_action.py loop shape (50 steps, reset + map_size
                           change every 10) driven by the one global MT19937 stream
"""
import json
import os
import sys
import zlib
from collections import deque

import numpy as np

import gym
import gym_novel_gridworlds  # noqa: F401  (registers the ids)
from gym_novel_gridworlds.novelty_wrappers import inject_novelty

OUT = os.path.dirname(os.path.abspath(__file__))
POGO = 'NovelGridworld-Pogostick-v1'
BOW = 'NovelGridworld-Bow-v1'
POGO0, BOW0 = 'NovelGridworld-Pogostick-v0', 'NovelGridworld-Bow-v0'

# name -> (env_id, map_size, novelty args or None)
CFGS = {
    'pogo10':   (POGO, 10, None),                                   # BASELINE config 2
    'bow20':    (BOW, 20, None),                                    # BASELINE config 3
    'axe10':    (POGO, 10, ('axe', 'medium', 'wooden', '')),        # BASELINE config 4
    'add32':    (POGO, 32, ('additem', 'hard', 'arrow', '')),       # BASELINE config 5
    # breadth (other sizes / argument variants of the same components)
    'pogo13':   (POGO, 13, None),
    'bow10':    (BOW, 10, None),
    'axe12bi':  (POGO, 12, ('axe', 'medium', 'iron', 'true')),
    'add12m':   (POGO, 12, ('additem', 'medium', 'spring', '')),
    'add11e':   (POGO, 11, ('additem', 'easy', 'arrow', '')),
    'bowaxe16': (BOW, 16, ('axe', 'medium', 'wooden', 'false')),
    'axeeasy10': (POGO, 10, ('axe', 'easy', 'wooden', '')),         # AxeEasy: the axe starts in the inventory
    # LUT-only novelties
    'brkinc10':    (POGO, 10, ('breakincrease', 'hard', '', '')),
    'brkinclog12': (POGO, 12, ('breakincrease', 'hard', 'tree_log', '')),
    'extdec10':    (BOW, 10, ('extractincdec', 'hard', 'decrease', '')),
    'axetbe10':    (POGO, 10, ('axetobreak', 'easy', 'wooden', '')),
    'axetbm12':    (BOW, 12, ('axetobreak', 'medium', 'iron', '')),
    'remape10':    (POGO, 10, ('remapaction', 'easy', '', '')),
    'remapm10':    (BOW, 10, ('remapaction', 'medium', '', '')),
    'remaph10':    (POGO, 10, ('remapaction', 'hard', '', '')),
    'chop10':      (POGO, 10, ('addchop', 'hard', '', '')),
    'jump12':      (BOW, 12, ('addjump', 'hard', '', '')),
    'axehard10':   (POGO, 10, ('axe', 'hard', 'wooden', '')),       # craftable axe, ingredients on the map
    'axehardi12':  (BOW, 12, ('axe', 'hard', 'iron', 'true')),      # new item 'iron' (add_new_items -> reset at injection)
    'atbhard10':   (POGO, 10, ('axetobreak', 'hard', 'wooden', '')),   # ingredients start in the inventory
    'atbhardi11':  (BOW, 11, ('axetobreak', 'hard', 'iron', '')),
    # reset-time map edits + step predicates
    'fence10e':    (POGO, 10, ('fence', 'easy', 'oak', '')),
    'fence12h':    (BOW, 12, ('fence', 'hard', 'jungle', '')),
    'fencer10e':   (POGO, 10, ('fencerestriction', 'easy', 'oak', '')),
    'fencer10m':   (POGO, 10, ('fencerestriction', 'medium', 'oak', '')),
    'fencer12h':   (BOW, 12, ('fencerestriction', 'hard', 'jungle', '')),
    'repl10m':     (POGO, 10, ('replaceitem', 'medium', 'tree_log', 'brick')),
    'replwall12e': (BOW, 12, ('replaceitem', 'easy', 'wall', 'brick')),
    'fire10h':     (POGO, 10, ('firewall', 'hard', '', '')),
    'fire14m':     (BOW, 14, ('firewall', 'medium', '', '')),
    'crate10m':    (POGO, 10, ('crate', 'medium', '', '')),
    'crate12h':    (BOW, 12, ('crate', 'hard', '', '')),
    'crate11e':    (POGO, 11, ('crate', 'easy', '', '')),
    # stacked novelties (wrapper over wrapper): who handles Break, and in which order the reset passes run, follows the nesting
    'stk_axe_bi10':   (POGO, 10, [('axe', 'medium', 'wooden', ''), ('breakincrease', 'hard', '', '')]),
    'stk_bi_axe10':   (POGO, 10, [('breakincrease', 'hard', 'tree_log', ''), ('axe', 'medium', 'iron', 'true')]),
    'stk_add_axe12':  (POGO, 12, [('additem', 'easy', 'arrow', ''), ('axe', 'easy', 'wooden', '')]),
    'stk_atb_bi11':   (BOW, 11, [('axetobreak', 'easy', 'iron', ''), ('breakincrease', 'hard', '', '')]),
    'stk_fen_fire12': (POGO, 12, [('fence', 'easy', 'oak', ''), ('firewall', 'medium', '', '')]),
    # (firewall then fence crashes in the reference: Fence.reset fences the fire_wall cells of the ring -> IndexError)
    'stk_add_repl12': (BOW, 12, [('additem', 'medium', 'arrow', ''), ('replaceitem', 'medium', 'arrow', 'dart')]),
    'stk_fire_axe10': (POGO, 10, [('firewall', 'medium', '', ''), ('axe', 'medium', 'wooden', '')]),     # fire check skipped on Break
    'stk_fire_axeh10': (POGO, 10, [('firewall', 'medium', '', ''), ('axe', 'hard', 'wooden', '')]),      # ... and on Craft_<axe>
    'stk_fr_axe10':   (POGO, 10, [('fencerestriction', 'hard', 'oak', ''), ('axe', 'easy', 'wooden', '')]),   # restriction never consulted
    'stk_axe_fr10':   (POGO, 10, [('axe', 'easy', 'wooden', ''), ('fencerestriction', 'medium', 'oak', '')]),
    'stk_crate_fr12': (BOW, 12, [('crate', 'medium', '', ''), ('fencerestriction', 'hard', 'oak', '')]),      # crate inside the restriction
    'stk_fr_crate12': (BOW, 12, [('fencerestriction', 'hard', 'oak', ''), ('crate', 'medium', '', '')]),
    'stk_crate_bi10': (POGO, 10, [('crate', 'hard', '', ''), ('breakincrease', 'hard', '', '')]),             # no crate bonus
    # two reset passes of the SAME kind in one stack (round 2): the reference nests them freely (novelty_wrappers.py:1013-1034, :1070-1076, :1129-1160)
    'stk_add_crate12': (POGO, 12, [('additem', 'medium', 'arrow', ''), ('crate', 'medium', '', '')]),
    'stk_crate_add12': (BOW, 12, [('crate', 'hard', '', ''), ('additem', 'easy', 'arrow', '')]),
    'stk_fen_fr12':    (POGO, 12, [('fence', 'easy', 'oak', ''), ('fencerestriction', 'hard', 'jungle', '')]),
    'stk_fr_fen12':    (BOW, 12, [('fencerestriction', 'medium', 'oak', ''), ('fence', 'medium', 'jungle', '')]),
    'stk_repl_fire12': (POGO, 12, [('replaceitem', 'medium', 'crafting_table', 'anvil'), ('firewall', 'hard', '', '')]),
    'stk_fire_repl12': (BOW, 12, [('firewall', 'medium', '', ''), ('replaceitem', 'hard', 'wool', 'silk')]),
    # the v0 variants
    'pogov0_10':   (POGO0, 10, None),
    'pogov0_14':   (POGO0, 14, ('axe', 'medium', 'wooden', '')),
    'bowv0_12':    (BOW0, 12, None),
}
REMAP_SEED = {'remape10': 11, 'remapm10': 12, 'remaph10': 13,      # np.random.seed right before inject_novelty
              'crate10m': 31, 'crate12h': 32, 'crate11e': 33, 'stk_crate_fr12': 34, 'stk_fr_crate12': 35, 'stk_crate_bi10': 36,
              'stk_add_crate12': 37, 'stk_crate_add12': 38}      # (remapaction shuffles / Crate draws its ingredients there)
DIRS = ['NORTH', 'SOUTH', 'WEST', 'EAST']


def make_env(cfg):
    env_id, S, nov = CFGS[cfg]
    env = gym.make(env_id)
    env.map_size = S            # on the BASE env, before wrapping
    if nov is not None:
        if cfg in REMAP_SEED:
            np.random.seed(REMAP_SEED[cfg])     # remapaction shuffles with the global stream at injection time
        for one in novelty_list(nov):           # a stack: injected in order, the last one is the outermost wrapper
            env = inject_novelty(env, *one)
    return env


def novelty_list(nov):
    return [] if nov is None else ([nov] if isinstance(nov[0], str) else list(nov))


def snap(base):
    ids = base.items_id
    inv = np.zeros(len(ids), np.int32)
    for name, q in base.inventory_items_quantity.items():
        inv[ids[name]] = q
    sel = ids[base.selected_item] if base.selected_item else 0
    return (np.asarray(base.map, dtype=np.int8).ravel().copy(),
            np.array(base.agent_location, np.int32), np.int32(base.agent_facing_id), np.int32(sel), inv)


def next_word():
    # legacy randint with rng == 0xFFFFFFFF returns the raw next 32-bit MT19937 output
    return int(np.random.randint(0, 2 ** 32, dtype=np.uint32))


class Strings:
    def __init__(self):
        self.idx = {}
        self.lst = []

    def __call__(self, s):
        if s not in self.idx:
            self.idx[s] = len(self.lst)
            self.lst.append(s)
        return self.idx[s]


def cost_pair(c):
    return float(c), int(isinstance(c, (int, np.integer)) and not isinstance(c, bool))


# ---------------------------------------------------------------- G1 spec
def spec_of(cfg):
    env = make_env(cfg)
    base = env.unwrapped
    np.random.seed(0)
    env.reset()
    d = {
        'env_id': base.env_id, 'map_size': int(base.map_size),
        'items_id': {k: int(v) for k, v in base.items_id.items()},
        'actions_id': {k: int(v) for k, v in base.actions_id.items()},
        'action_space_n': int(env.action_space.n),
        'base_action_space_n': int(base.action_space.n),
        'recipes': {k: {'input': [[i, int(q)] for i, q in v['input'].items()],
                        'output': [[i, int(q)] for i, q in v['output'].items()]}
                    for k, v in base.recipes.items()},
        'items_quantity': [[k, int(v)] for k, v in base.items_quantity.items()],
        'entities': sorted(base.entities), 'unbreakable_items': sorted(base.unbreakable_items),
        'goal_item_to_craft': base.goal_item_to_craft,
        'reward_intermediate': base.reward_intermediate, 'reward_done': base.reward_done,
        'novelty': CFGS[cfg][2],
    }
    w = env
    while hasattr(w, 'env'):
        if 'crate_ingredients' in vars(w):
            d['crate_ingredients'] = [str(x) for x in w.crate_ingredients]
        w = w.env
    A = len(base.actions_id)
    errs = []
    for a in (A, A + 5, -1):
        try:
            env.step(a)
            errs.append([a, None, None])
        except Exception as e:  # noqa: BLE001
            errs.append([a, type(e).__name__, str(e)])
    d['invalid_action_errors'] = errs
    return d


def novelty_arg_errors():
    cases = [('axe', 'medium', '', ''), ('axe', 'medium', 'gold', ''), ('axe', 'medium', 'wooden', 'maybe'),
             ('axe', 'extreme', 'wooden', ''), ('additem', 'hard', '', ''), ('additem', 'harder', 'arrow', ''),
             ('teleport', 'hard', '', '')]
    out = []
    for c in cases:
        env = gym.make(POGO)
        try:
            inject_novelty(env, *c)
            out.append([list(c), None, None])
        except Exception as e:  # noqa: BLE001
            out.append([list(c), type(e).__name__, str(e)])
    return out


def novelty_arg_errors2():
    """(env id, args) -> exception, for the row-2 novelties."""
    cases = [(POGO, ('breakincrease', 'hard', 'unobtainium', '')), (POGO, ('extractincdec', 'hard', '', '')),
             (POGO, ('extractincdec', 'hard', 'increase', '')), (BOW, ('extractincdec', 'hard', 'increase', '')),
             (BOW, ('extractincdec', 'hard', 'sideways', '')), (POGO, ('axetobreak', 'easy', 'gold', '')),
             (POGO, ('axetobreak', 'nope', 'wooden', '')), (POGO, ('remapaction', 'nope', '', ''))]
    out = []
    for env_id, c in cases:
        env = gym.make(env_id)
        try:
            inject_novelty(env, *c)
            out.append([env_id, list(c), None, None])
        except Exception as e:  # noqa: BLE001
            out.append([env_id, list(c), type(e).__name__, str(e)])
    return out


def exhaustion_cases():
    out = []
    for env_id in (POGO, BOW):
        for S in (5, 6, 7, 8):
            for seed in range(12):
                env = gym.make(env_id)
                env.map_size = S
                np.random.seed(seed)
                try:
                    env.reset()
                    m, loc, f, _, _ = snap(env)
                    out.append({'env_id': env_id, 'S': S, 'seed': seed, 'ok': True, 'crc': zlib.crc32(m.tobytes()),
                                'loc': loc.tolist(), 'facing': int(f), 'next_word': next_word()})
                except AssertionError as e:
                    out.append({'env_id': env_id, 'S': S, 'seed': seed, 'ok': False, 'error': str(e)})
    return out


# ---------------------------------------------------------------- G2 resets
def gen_resets(cfg, nseeds, out):
    env = make_env(cfg)
    base = env.unwrapped
    maps, locs, facs, words, invs = [], [], [], [], []
    for seed in range(nseeds):
        np.random.seed(seed)
        for _ in range(3):
            env.reset()
            m, loc, f, sel, inv = snap(base)
            assert sel == 0 and (not inv.any() or cfg in ('axeeasy10', 'axetbe10', 'atbhard10', 'atbhardi11', 'stk_add_axe12', 'stk_atb_bi11', 'stk_fr_axe10', 'stk_axe_fr10'))
            maps.append(m), locs.append(loc), facs.append(f), invs.append(inv)
        words.append(next_word())
    out['rs_map'] = np.array(maps, np.int8).reshape(nseeds, 3, -1)
    out['rs_loc'] = np.array(locs, np.int32).reshape(nseeds, 3, 2)
    out['rs_facing'] = np.array(facs, np.int32).reshape(nseeds, 3)
    out['rs_next_word'] = np.array(words, np.uint32)
    out['rs_inv'] = np.array(invs, np.int32).reshape(nseeds, 3, -1)


# ---------------------------------------------------------------- G3 traces
def gen_trace(cfg, k, T, out, strings):
    env = make_env(cfg)
    base = env.unwrapped
    ids = base.items_id
    K, A = len(ids), len(base.actions_id)
    rs = np.random.RandomState(7919 * (k + 1))
    inj_pool = [n for n in ids if n not in ('air', 'wall', base.goal_item_to_craft)]
    goal_in = base.recipes[base.goal_item_to_craft]['input']

    acts = np.zeros(T, np.int32)
    rew, sc = np.zeros(T, np.int32), np.zeros(T, np.int32)
    done, res, cint = np.zeros(T, np.uint8), np.zeros(T, np.uint8), np.zeros(T, np.uint8)
    cost = np.zeros(T, np.float64)
    msg = np.zeros(T, np.int32)
    loc, fac, sel, inv = np.zeros((T, 2), np.int32), np.zeros(T, np.int32), np.zeros(T, np.int32), np.zeros((T, K), np.int32)
    md_t, md_i, md_v = [], [], []
    rl_t, rl_map, rl_loc, rl_fac, rl_inv = [], [], [], [], []
    inj_t, inj_item, inj_q = [], [], []

    np.random.seed(1000 + k)
    need_reset, done_run = True, 0
    prev_map = None
    for t in range(T):
        if need_reset:
            env.reset()
            m, l, f, _, iv0 = snap(base)
            rl_t.append(t), rl_map.append(m), rl_loc.append(l), rl_fac.append(f), rl_inv.append(iv0)
            prev_map, need_reset, done_run = m, False, 0
        if t % 37 == 36:
            for name in rs.choice(inj_pool, size=3, replace=False):
                q = int(rs.randint(0, 6))
                if q:
                    base.inventory_items_quantity[name] += q
                    inj_t.append(t), inj_item.append(ids[name]), inj_q.append(q)
        if t % 501 == 500:
            for name, q in goal_in.items():
                base.inventory_items_quantity[name] += q
                inj_t.append(t), inj_item.append(ids[name]), inj_q.append(q)
        a = int(rs.randint(A))
        acts[t] = a
        _, r, d, info = env.step(a)
        m, l, f, s, iv = snap(base)
        rew[t], done[t], res[t] = r, d, info['result']
        cost[t], cint[t] = cost_pair(info['step_cost'])
        msg[t] = strings(info['message'])
        loc[t], fac[t], sel[t], inv[t], sc[t] = l, f, s, iv, base.step_count
        ch = np.nonzero(m != prev_map)[0]
        for i in ch:
            md_t.append(t), md_i.append(i), md_v.append(m[i])
        prev_map = m
        if d:
            done_run += 1
            if done_run >= 25:      # 25 sticky-done steps, then a fresh episode
                need_reset = True
    p = 'tr%d_' % k
    out[p + 'action'], out[p + 'reward'], out[p + 'done'], out[p + 'result'] = acts, rew, done, res
    out[p + 'cost'], out[p + 'cost_is_int'], out[p + 'msg'] = cost, cint, msg
    out[p + 'loc'], out[p + 'facing'], out[p + 'sel'], out[p + 'inv'], out[p + 'step_count'] = loc, fac, sel, inv, sc
    out[p + 'md_t'], out[p + 'md_i'], out[p + 'md_v'] = (np.array(md_t, np.int32), np.array(md_i, np.int32),
                                                         np.array(md_v, np.int8))
    out[p + 'rl_t'], out[p + 'rl_map'] = np.array(rl_t, np.int32), np.array(rl_map, np.int8)
    out[p + 'rl_loc'], out[p + 'rl_facing'] = np.array(rl_loc, np.int32), np.array(rl_fac, np.int32)
    out[p + 'rl_inv'] = np.array(rl_inv, np.int32)
    out[p + 'inj_t'], out[p + 'inj_item'], out[p + 'inj_q'] = (np.array(inj_t, np.int32), np.array(inj_item, np.int32),
                                                               np.array(inj_q, np.int32))
    return int(done.sum())


# ---------------------------------------------------------------- G4 single steps from injected states
def inject_state(base, m, loc, facing, sel, inv):
    S = base.map_size
    names = {v: k for k, v in base.items_id.items()}
    base.map = np.array(m, dtype=int).reshape(S, S)
    base.agent_location = (int(loc[0]), int(loc[1]))
    base.set_agent_facing(DIRS[int(facing)])
    base.inventory_items_quantity = {names[i]: int(inv[i]) for i in range(len(inv))}
    base.selected_item = names[int(sel)] if sel else ''
    base.update_block_in_front()          # keep the cached front block coherent, as reset/step do


def gen_single_steps(cfg, n, out, strings):
    env = make_env(cfg)
    base = env.unwrapped
    np.random.seed(1)
    env.reset()
    ids = base.items_id
    S, K, A = base.map_size, len(ids), len(base.actions_id)
    wall, goal = ids['wall'], ids[base.goal_item_to_craft]
    placeable = [i for i in range(1, K) if i != wall]
    selectable = [ids[a.split('_', 1)[1]] for a in base.actions_id if a.startswith('Select_')]
    axe = [ids[n_] for n_ in ids if n_.endswith('_axe')]
    rs = np.random.RandomState(424242)

    pm = np.zeros((n, S * S), np.int8)
    ploc, pfac, psel, pinv = np.zeros((n, 2), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, K), np.int32)
    act = np.zeros(n, np.int32)
    qloc, qfac, qsel, qinv = np.zeros((n, 2), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, K), np.int32)
    rew, done, res = np.zeros(n, np.int32), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    cost, cint, msg = np.zeros(n, np.float64), np.zeros(n, np.uint8), np.zeros(n, np.int32)
    md_c, md_i, md_v = [], [], []
    for c in range(n):
        m = np.zeros((S, S), np.int8)
        m[0, :] = m[-1, :] = m[:, 0] = m[:, -1] = wall
        dens = rs.choice([0.1, 0.45, 0.8])
        fill = rs.rand(S - 2, S - 2) < dens
        vals = rs.choice(placeable + ([wall] if rs.rand() < 0.2 else []), size=(S - 2, S - 2))
        m[1:-1, 1:-1] = np.where(fill, vals, 0)
        loc = rs.randint(1, S - 1, size=2)
        m[loc[0], loc[1]] = 0
        f = rs.randint(4)
        prof = rs.rand()
        if prof < 0.2:
            iv = np.zeros(K, np.int32)
        elif prof < 0.7:
            iv = rs.randint(0, 7, size=K).astype(np.int32)
        elif prof < 0.93:
            iv = np.full(K, 10, np.int32)
        else:
            iv = rs.randint(0, 7, size=K).astype(np.int32)
        iv[0] = iv[wall] = 0
        iv[goal] = 1 + rs.randint(2) if prof >= 0.93 else 0
        s = 0
        if rs.rand() < 0.6:
            s = int(rs.choice(axe)) if (axe and rs.rand() < 0.6) else int(rs.choice(selectable))
        a = int(rs.randint(A))
        inject_state(base, m.ravel(), loc, f, s, iv)
        _, r, d, info = env.step(a)
        m2, l2, f2, s2, iv2 = snap(base)
        pm[c], ploc[c], pfac[c], psel[c], pinv[c], act[c] = m.ravel(), loc, f, s, iv, a
        qloc[c], qfac[c], qsel[c], qinv[c] = l2, f2, s2, iv2
        rew[c], done[c], res[c] = r, d, info['result']
        cost[c], cint[c] = cost_pair(info['step_cost'])
        msg[c] = strings(info['message'])
        for i in np.nonzero(m2 != m.ravel())[0]:
            md_c.append(c), md_i.append(i), md_v.append(m2[i])
    for k_, v in dict(pre_map=pm, pre_loc=ploc, pre_facing=pfac, pre_sel=psel, pre_inv=pinv, action=act,
                      post_loc=qloc, post_facing=qfac, post_sel=qsel, post_inv=qinv, reward=rew, done=done,
                      result=res, cost=cost, cost_is_int=cint, msg=msg, md_c=np.array(md_c, np.int32),
                      md_i=np.array(md_i, np.int32), md_v=np.array(md_v, np.int8)).items():
        out['ss_' + k_] = v


# ---------------------------------------------------------------- G5 scripted solver
DR, DC = [-1, 1, 0, 0], [0, 0, -1, 1]
LEFT, RIGHT = [2, 3, 1, 0], [3, 2, 0, 1]


def plan_to(base, goal_fn):
    """BFS over (r,c,facing) with Forward/Left/Right; goal_fn(r,c,f)->bool. Returns action list or None."""
    m = base.map
    start = (base.agent_location[0], base.agent_location[1], base.agent_facing_id)
    prev = {start: None}
    dq = deque([start])
    while dq:
        st = dq.popleft()
        r, c, f = st
        if goal_fn(r, c, f):
            path = []
            while prev[st] is not None:
                st, a = prev[st]
                path.append(a)
            return path[::-1]
        nxt = [((r, c, LEFT[f]), 1), ((r, c, RIGHT[f]), 2)]
        fr, fc = r + DR[f], c + DC[f]
        if m[fr][fc] == 0:
            nxt.append(((fr, fc, f), 0))
        for ns, a in nxt:
            if ns not in prev:
                prev[ns] = (st, a)
                dq.append(ns)
    return None


def facing_item(base, item_id):
    m = base.map
    return lambda r, c, f: m[r + DR[f]][c + DC[f]] == item_id


def solve(env, rec):
    """Drives the reference to done; rec(action) performs env.step and records. Returns True if done."""
    base = env.unwrapped
    ids, A = base.items_id, base.actions_id
    inv = base.inventory_items_quantity

    def go(goal_fn):
        p = plan_to(base, goal_fn)
        if p is None:
            return False
        for a in p:
            rec(a)
        return True

    has_axe = [n for n in ids if n.endswith('_axe')]
    if has_axe and has_axe[0] in base.entities:
        axe_id = ids[has_axe[0]]
        m = base.map
        if (m == axe_id).any():
            # walking into the 3x3 neighbourhood picks the axe up (grab runs after every action)
            if not go(lambda r, c, f: any(m[r + a][c + b] == axe_id for a in (-1, 0, 1) for b in (-1, 0, 1))):
                return False
            if inv[has_axe[0]] < 1:
                rec(1)      # already in range at reset: any step triggers the pick-up
        if inv[has_axe[0]] >= 1:
            rec(A['Select_' + has_axe[0]])
    if base.env_id == POGO:
        for _ in range(3):
            if not go(facing_item(base, ids['tree_log'])):
                return False
            rec(A['Break'])
        for _ in range(3):
            rec(A['Craft_plank'])
        rec(A['Craft_stick']), rec(A['Craft_stick'])
        if not go(facing_item(base, ids['crafting_table'])):
            return False
        rec(A['Craft_tree_tap'])
        m = base.map
        log = ids['tree_log']

        def tap_spot(r, c, f):
            fr, fc = r + DR[f], c + DC[f]
            if m[fr][fc] != 0:
                return False
            return any(m[fr + DR[d]][fc + DC[d]] == log for d in range(4))
        if not go(tap_spot):
            return False
        rec(A['Place_tree_tap'])
        rec(A['Extract_rubber'])
        if not go(facing_item(base, ids['crafting_table'])):
            return False
        rec(A['Craft_pogo_stick'])
    else:
        for _ in range(2):
            if not go(facing_item(base, ids['tree_log'])):
                return False
            rec(A['Break'])
        if not go(facing_item(base, ids['wool'])):
            return False
        rec(A['Extract_string'])
        rec(A['Craft_plank'])
        rec(A['Craft_stick'])
        if not go(facing_item(base, ids['crafting_table'])):
            return False
        rec(A['Craft_bow'])
    for a in (0, 1, 3, A['Craft_plank'], 2):      # sticky-done tail
        rec(a)
    return inv[base.goal_item_to_craft] >= 1


def gen_solved(cfg, nep, out, strings):
    got = 0
    seed = 0
    while got < nep and seed < 200:
        env = make_env(cfg)
        base = env.unwrapped
        np.random.seed(500 + seed)
        seed += 1
        env.reset()
        m0, l0, f0, _, i0 = snap(base)
        K = len(base.items_id)
        rows = []
        prev = [m0]
        diffs = []

        def rec(a):
            _, r, d, info = env.step(a)
            m, l, f, s, iv = snap(base)
            t = len(rows)
            for i in np.nonzero(m != prev[0])[0]:
                diffs.append((t, i, m[i]))
            prev[0] = m
            c, ci = cost_pair(info['step_cost'])
            rows.append((a, r, d, info['result'], c, ci, strings(info['message']), l, f, s, iv))
        if not solve(env, rec):
            continue
        p = 'so%d_' % got
        got += 1
        out[p + 'map0'], out[p + 'loc0'], out[p + 'facing0'], out[p + 'inv0'] = m0, l0, f0, i0
        out[p + 'action'] = np.array([r[0] for r in rows], np.int32)
        out[p + 'reward'] = np.array([r[1] for r in rows], np.int32)
        out[p + 'done'] = np.array([r[2] for r in rows], np.uint8)
        out[p + 'result'] = np.array([r[3] for r in rows], np.uint8)
        out[p + 'cost'] = np.array([r[4] for r in rows], np.float64)
        out[p + 'cost_is_int'] = np.array([r[5] for r in rows], np.uint8)
        out[p + 'msg'] = np.array([r[6] for r in rows], np.int32)
        out[p + 'loc'] = np.array([r[7] for r in rows], np.int32)
        out[p + 'facing'] = np.array([r[8] for r in rows], np.int32)
        out[p + 'sel'] = np.array([r[9] for r in rows], np.int32)
        out[p + 'inv'] = np.array([r[10] for r in rows], np.int32).reshape(len(rows), K)
        out[p + 'md_t'] = np.array([d[0] for d in diffs], np.int32)
        out[p + 'md_i'] = np.array([d[1] for d in diffs], np.int32)
        out[p + 'md_v'] = np.array([d[2] for d in diffs], np.int8)
    return got


# ---------------------------------------------------------------- C1 random_action.py loop shape
def gen_c1loop(nseeds, out, strings):
    for k in range(nseeds):
        env = gym.make(POGO)
        base = env.unwrapped
        A = env.action_space.n
        np.random.seed(k)
        env.reset()
        sizes, maps = [base.map_size], [snap(base)[0]]
        rows = []
        for i in range(50):
            a = int(np.random.randint(A))           # Discrete.sample() of the stand-in: global stream
            _, r, d, info = env.step(a)
            m, l, f, s, iv = snap(base)
            c, ci = cost_pair(info['step_cost'])
            rows.append((a, r, d, info['result'], c, ci, strings(info['message']), l, f, s, iv, zlib.crc32(m.tobytes())))
            if (i + 1) % 10 == 0:
                base.map_size = int(np.random.randint(low=10, high=20, size=1)[0])
                env.reset()
                sizes.append(base.map_size), maps.append(snap(base)[0])
        p = 'c%d_' % k
        out[p + 'sizes'] = np.array(sizes, np.int32)
        for j, m in enumerate(maps):
            out[p + 'map%d' % j] = m
        out[p + 'action'] = np.array([r[0] for r in rows], np.int32)
        out[p + 'reward'] = np.array([r[1] for r in rows], np.int32)
        out[p + 'done'] = np.array([r[2] for r in rows], np.uint8)
        out[p + 'result'] = np.array([r[3] for r in rows], np.uint8)
        out[p + 'cost'] = np.array([r[4] for r in rows], np.float64)
        out[p + 'cost_is_int'] = np.array([r[5] for r in rows], np.uint8)
        out[p + 'msg'] = np.array([r[6] for r in rows], np.int32)
        out[p + 'loc'] = np.array([r[7] for r in rows], np.int32)
        out[p + 'facing'] = np.array([r[8] for r in rows], np.int32)
        out[p + 'sel'] = np.array([r[9] for r in rows], np.int32)
        out[p + 'inv'] = np.array([r[10] for r in rows], np.int32)
        out[p + 'crc'] = np.array([r[11] for r in rows], np.uint32)
        out[p + 'next_word'] = np.array([next_word()], np.uint32)


# ---------------------------------------------------------------- LimitActions (wrappers.py:57-85), with and without remapaction on top
LIMITED = ['Forward', 'Left', 'Right', 'Break', 'Craft_plank', 'Craft_stick', 'Select_tree_log']


def gen_limit(out, strings, spec):
    from gym_novel_gridworlds.wrappers import LimitActions
    meta = {}
    for name, remap_seed in (('plain', None), ('remapped', 21)):
        env = gym.make(POGO)
        env = LimitActions(env, set(LIMITED))
        if remap_seed is not None:
            np.random.seed(remap_seed)
            env = inject_novelty(env, 'remapaction', 'hard')
        base = env.unwrapped
        np.random.seed(77)
        env.reset()
        m0, l0, f0, _, _ = snap(base)
        rs = np.random.RandomState(5)
        rows = []
        n = len(env.limited_actions_id)
        for t in range(600):
            if t % 50 == 49:
                base.inventory_items_quantity['tree_log'] += 2
                base.inventory_items_quantity['plank'] += 2
            a = int(rs.randint(n))
            _, r, d, info = env.step(a)
            m, l, f, s, iv = snap(base)
            c, ci = cost_pair(info['step_cost'])
            rows.append((a, r, d, info['result'], c, ci, strings(info['message']), l, f, s, iv, zlib.crc32(m.tobytes())))
        p = 'lim_%s_' % name
        out[p + 'map0'], out[p + 'loc0'], out[p + 'facing0'] = m0, l0, f0
        for j, key in enumerate(['action', 'reward', 'done', 'result', 'cost', 'cost_is_int', 'msg', 'loc', 'facing', 'sel', 'inv', 'crc']):
            out[p + key] = np.array([r[j] for r in rows])
        errs = []
        for a in (n, -1):
            try:
                env.step(a)
                errs.append([a, None, None])
            except Exception as e:  # noqa: BLE001
                errs.append([a, type(e).__name__, str(e)])
        meta[name] = {'limited_actions_id': {k: int(v) for k, v in env.limited_actions_id.items()},
                      'action_space_n': int(env.action_space.n), 'errors': errs}
    spec['limit_actions'] = meta


# ---------------------------------------------------------------- main
PLAN = {  # cfg: (reset seeds, traces, steps per trace, single-step cases, solved episodes)
    'pogo10': (48, 6, 1600, 6000, 4), 'bow20': (32, 4, 1600, 3000, 3), 'axe10': (48, 6, 1600, 6000, 4),
    'add32': (24, 3, 1200, 500, 0), 'pogo13': (16, 2, 1000, 1500, 2), 'bow10': (16, 2, 1000, 3000, 2),
    'axe12bi': (16, 3, 1200, 4000, 3), 'add12m': (16, 2, 1000, 1500, 0), 'add11e': (16, 1, 600, 500, 0),
    'bowaxe16': (16, 2, 1000, 2500, 2), 'axeeasy10': (16, 3, 1200, 3000, 2),
    'brkinc10': (8, 2, 1000, 3000, 2), 'brkinclog12': (8, 2, 800, 2000, 1), 'extdec10': (8, 2, 1000, 3000, 2),
    'axetbe10': (8, 2, 1000, 3000, 2), 'axetbm12': (8, 2, 1000, 3000, 2), 'remape10': (8, 2, 1000, 2500, 2),
    'remapm10': (8, 2, 1000, 2500, 2), 'remaph10': (8, 2, 1000, 2500, 2), 'chop10': (8, 2, 1000, 3000, 1),
    'jump12': (8, 2, 1000, 3000, 1), 'pogov0_10': (32, 3, 1200, 4000, 0), 'pogov0_14': (16, 2, 1000, 2500, 0),
    'bowv0_12': (16, 2, 1000, 3000, 0),
    'axehard10': (12, 2, 1200, 4000, 2), 'axehardi12': (8, 2, 1000, 3000, 1), 'atbhard10': (12, 2, 1200, 4000, 0),
    'atbhardi11': (8, 2, 1000, 3000, 0),
    'fence10e': (16, 2, 1000, 2500, 0), 'fence12h': (12, 2, 800, 2000, 0), 'fencer10e': (8, 1, 800, 2000, 0),
    'fencer10m': (16, 3, 1200, 5000, 0), 'fencer12h': (16, 3, 1200, 5000, 0), 'repl10m': (16, 2, 1000, 2500, 0),
    'replwall12e': (16, 2, 1000, 2500, 1), 'fire10h': (16, 3, 1200, 4000, 1), 'fire14m': (16, 3, 1200, 4000, 1),
    'crate10m': (12, 3, 1200, 4000, 2), 'crate12h': (12, 2, 1000, 3000, 1), 'crate11e': (8, 2, 1000, 2500, 1),
    'stk_axe_bi10': (8, 2, 1000, 3000, 0), 'stk_bi_axe10': (8, 2, 1000, 3000, 0), 'stk_add_axe12': (8, 2, 800, 2000, 0),
    'stk_atb_bi11': (8, 2, 1000, 3000, 0), 'stk_fen_fire12': (12, 2, 800, 2000, 0),
    'stk_add_repl12': (12, 2, 800, 2000, 0),
    'stk_fire_axe10': (8, 2, 1000, 4000, 0), 'stk_fire_axeh10': (8, 2, 1000, 4000, 0), 'stk_fr_axe10': (8, 2, 800, 4000, 0),
    'stk_axe_fr10': (8, 2, 800, 4000, 0), 'stk_crate_fr12': (8, 2, 800, 4000, 0), 'stk_fr_crate12': (8, 2, 800, 4000, 0),
    'stk_crate_bi10': (8, 2, 800, 3000, 0),
    'stk_add_crate12': (12, 2, 800, 3000, 0), 'stk_crate_add12': (12, 2, 800, 3000, 0), 'stk_fen_fr12': (12, 2, 800, 4000, 0),
    'stk_fr_fen12': (12, 2, 800, 4000, 0), 'stk_repl_fire12': (12, 2, 800, 2000, 0), 'stk_fire_repl12': (12, 2, 800, 2000, 0),
}


CHECK = {'on': False, 'bad': []}


def emit(name, out):
    """Write a fixture - or, with --check, compare what was just generated with the committed file: same keys, same dtypes,
    shapes and values (a hand-edited or stale .npz shows up here)."""
    path = os.path.join(OUT, name + '.npz')
    if not CHECK['on']:
        np.savez_compressed(path, **out)
        return
    try:
        old = dict(np.load(path))
    except OSError as e:
        CHECK['bad'].append('%s: %s' % (name, e))
        return
    for k in sorted(set(out) | set(old)):
        if k not in old or k not in out:
            CHECK['bad'].append('%s: key %s only in the %s' % (name, k, 'generated data' if k in out else 'committed file'))
            continue
        a, b = np.asarray(out[k]), old[k]
        if a.dtype != b.dtype or a.shape != b.shape or not np.array_equal(a, b):
            CHECK['bad'].append('%s: %s differs (%s %s vs committed %s %s)' % (name, k, a.dtype, a.shape, b.dtype, b.shape))


def main():
    """gen_golden.py [--check] [cfg ...]: regenerate (or, with --check, verify in memory) the fixtures of the named
    configurations, or of all of them."""
    args = sys.argv[1:]
    CHECK['on'] = '--check' in args
    only = [a for a in args if a != '--check']
    strings = Strings()
    sfile = os.path.join(OUT, 'spec.json')
    spec = json.load(open(sfile)) if (only and os.path.exists(sfile)) else {}
    if only and 'messages' in spec:
        for s in spec['messages']:
            strings(s)
    spec.setdefault('cfgs', {})
    summary = {}
    for cfg, (nrs, ntr, T, nss, nso) in PLAN.items():
        if only and cfg not in only:
            continue
        out = {}
        spec['cfgs'][cfg] = spec_of(cfg)
        gen_resets(cfg, nrs, out)
        dones = sum(gen_trace(cfg, k, T, out, strings) for k in range(ntr))
        gen_single_steps(cfg, nss, out, strings)
        got = gen_solved(cfg, nso, out, strings) if nso else 0
        spec['cfgs'][cfg].update(n_reset_seeds=nrs, n_traces=ntr, trace_len=T, n_single=nss, n_solved=got)
        emit(cfg, out)
        summary[cfg] = dict(done_steps_in_traces=dones, solved=got,
                            bytes=os.path.getsize(os.path.join(OUT, cfg + '.npz')) if os.path.exists(os.path.join(OUT, cfg + '.npz')) else 0)
        print(cfg, summary[cfg], flush=True)
    if not only or 'c1loop' in only:
        out = {}
        gen_c1loop(6, out, strings)
        emit('c1loop', out)
        spec['novelty_arg_errors'] = novelty_arg_errors()
        spec['novelty_arg_errors2'] = novelty_arg_errors2()
        lim = {}
        gen_limit(lim, strings, spec)
        emit('limit', lim)
        spec['exhaustion'] = exhaustion_cases()
    spec['messages'] = strings.lst
    spec['generator'] = {'numpy': np.__version__, 'python': sys.version.split()[0],
                         'reference': 'gtatiya/gym-novel-gridworlds v1.2 (setup.py) at /root/reference'}
    if CHECK['on']:
        committed = json.load(open(sfile))
        for k in sorted(set(spec) | set(committed)):
            if k in ('generator',):
                continue
            if k == 'cfgs':
                for c in spec['cfgs']:
                    if json.dumps(spec['cfgs'][c], sort_keys=True) != json.dumps(committed['cfgs'].get(c), sort_keys=True):
                        CHECK['bad'].append('spec.json: cfgs[%s] differs' % c)
            elif not only and json.dumps(spec.get(k), sort_keys=True) != json.dumps(committed.get(k), sort_keys=True):
                CHECK['bad'].append('spec.json: %s differs' % k)
        if CHECK['bad']:
            print('FIXTURE CHECK FAILED:\n  ' + '\n  '.join(CHECK['bad'][:40]))
            sys.exit(1)
        print('fixture check ok: %s' % (', '.join(only) if only else 'every configuration'))
        return
    json.dump(spec, open(sfile, 'w'), indent=1, sort_keys=True)
    print('messages:', len(strings.lst))


if __name__ == '__main__':
    main()
