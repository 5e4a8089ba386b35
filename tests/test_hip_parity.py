"""Parity of the HIP path (through the C-ABI) against the golden vectors and the CPU oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest

import ngw_testlib as T
from gym_novel_gridworlds_amd import VecNovelGridworld
from gym_novel_gridworlds_amd.spec import make_spec
from oracle.ngw_oracle import Oracle

pytestmark = pytest.mark.gpu
ALL = [c for c in T.CFGS if c not in T.NO_FIXTURES]
STATE_KEYS = ('map', 'loc', 'facing', 'inv', 'selected', 'step_count', 'episode')


def oracle_state(o):
    st = o.st
    return dict(map=st.map, loc=st.loc, facing=st.facing, inv=st.inv, selected=st.selected, step_count=st.step_count,
                episode=st.episode)


def assert_state_equal(v, o, where):
    hs, os_ = v.get_state(), oracle_state(o)
    for k in STATE_KEYS:
        bad = np.nonzero((hs[k] != os_[k]).reshape(len(hs[k]), -1).any(1))[0]
        assert bad.size == 0, "%s: %s differs for %d envs, first env %d" % (where, k, bad.size, bad[0])


# ------------------------------------------------------------------ golden vectors from the reference
@pytest.mark.parametrize('cfg', ALL)
def test_traces_match_reference(cfg):
    assert T.replay_traces(cfg, T.HipBackend) > 0


@pytest.mark.parametrize('cfg', ALL)
def test_single_steps_match_reference(cfg):
    assert T.replay_single_steps(cfg, T.HipBackend) > 0


@pytest.mark.parametrize('cfg', [c for c in ALL if T.spec_json()['cfgs'][c]['n_solved']])
def test_solved_episodes_match_reference(cfg):
    assert T.replay_solved(cfg, T.HipBackend) > 0


# ------------------------------------------------------------------ oracle on the same seeded inputs
@pytest.mark.parametrize('cfg,n', [('pogo10', 5000), ('bow20', 1500), ('axe10', 4096), ('add32', 300), ('pogo13', 777),
                                   ('bow10', 1000), ('axe12bi', 1000), ('add12m', 640), ('add11e', 500), ('bowaxe16', 900),
                                   ('axeeasy10', 700), ('pogov0_10', 2000), ('pogov0_14', 600), ('bowv0_12', 600), ('axetbm12', 500),
                                   ('chop10', 300), ('axehard10', 800), ('axehardi12', 500), ('atbhard10', 500), ('fence10e', 1500),
                                   ('fence12h', 700), ('fencer10m', 1000), ('fencer12h', 600), ('repl10m', 1000), ('replwall12e', 800),
                                   ('fire10h', 1500), ('fire14m', 600), ('crate10m', 1000), ('crate12h', 600), ('fire32m', 200),
                                   ('fencer24h', 333), ('repl40e', 130), ('stk_fen_fire12', 700), ('stk_add_repl12', 600),
                                   ('stk_add_axe12', 500), ('stk_add_crate12', 600), ('stk_crate_add12', 500), ('stk_fen_fr12', 600),
                                   ('stk_fr_fen12', 500), ('stk_repl_fire12', 600), ('stk_fire_repl12', 500)])
def test_reset_matches_oracle(cfg, n):
    """reset(): template + per-env Philox item scatter (+ AddItem pass), three episodes, ragged N, masked reset."""
    spec = T.build_spec(cfg)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=99, env_index_base=12345)
    o = Oracle(spec.compile(), n, seed=99, env_index_base=12345)
    for ep in range(2):
        v.reset()
        assert o.reset() == 0
        assert_state_equal(v, o, '%s reset %d' % (cfg, ep))
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    v.reset(mask)
    o.reset(mask)
    assert_state_equal(v, o, cfg + ' masked reset')
    obs = v.get_observation()
    assert (obs['map'].reshape(n, -1) == o.st.map).all() and (obs['agent_location'] == o.st.loc).all()
    assert (obs['agent_facing_id'] == o.st.facing).all() and (obs['inventory_items_quantity'] == o.st.inv).all()


@pytest.mark.parametrize('cfg,n,steps,horizon', [('pogo10', 4096, 260, 50), ('bow20', 1024, 150, 40), ('axe10', 4096, 260, 50),
                                                 ('add32', 256, 60, 25), ('axe12bi', 1000, 120, 30), ('bow10', 999, 150, 0),
                                                 ('fencer10m', 2048, 200, 40), ('fencer12h', 777, 120, 30), ('fire10h', 4096, 200, 50),
                                                 ('fire14m', 500, 100, 0), ('crate10m', 2048, 200, 40), ('repl10m', 1000, 120, 30),
                                                 ('atbhard10', 1000, 150, 35), ('axehardi12', 640, 100, 30), ('fence12h', 500, 80, 25),
                                                 ('fire32m', 128, 60, 20), ('fencer24h', 256, 60, 20), ('stk_fen_fire12', 1000, 100, 25),
                                                 ('stk_bi_axe10', 1000, 100, 30), ('stk_atb_bi11', 800, 100, 0), ('stk_fire_axe10', 1500, 120, 40),
                                                 ('stk_crate_fr12', 800, 100, 30), ('stk_fr_crate12', 800, 100, 30), ('stk_fr_axe10', 800, 80, 25),
                                                 ('stk_add_crate12', 800, 80, 25), ('stk_fen_fr12', 800, 80, 25), ('stk_repl_fire12', 800, 80, 25),
                                                 ('stk_fire_repl12', 640, 100, 0)])
def test_autoreset_steps_match_oracle(cfg, n, steps, horizon):
    """Random actions with same-step autoreset (done or horizon): outputs every step, full state at checkpoints."""
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=7, autoreset=True, horizon=horizon)
    o = Oracle(spec.compile(), n, seed=7, autoreset=True, horizon=horizon)
    v.reset()
    o.reset()
    rs = np.random.RandomState(3)
    goal_in = spec.recipes[spec.goal_item_to_craft]['input']
    for t in range(steps):
        if t % 20 == 7:      # hand every 5th env the goal recipe's inputs so that some episodes end with done
            st = v.get_state()
            for name, q in goal_in.items():
                st['inv'][::5, spec.items_id[name]] += q
                o.st.inv[::5, spec.items_id[name]] += q
            v.set_state(0, inv=st['inv'])
        a = rs.randint(0, A, size=n).astype(np.int32)
        _, reward, done, info = v.step(a)
        assert o.step(a) == 0
        where = '%s step %d' % (cfg, t)
        assert (reward == o.reward).all(), where
        assert (done == o.done.astype(bool)).all(), where
        assert (info['result'] == o.result.astype(bool)).all() and (info['step_cost_code'] == o.cost_code).all(), where
        assert (info['message_code'] == o.msg_code).all() and (info['message_arg'] == o.msg_arg).all(), where
        if t % 25 == 24 or t == steps - 1:
            assert_state_equal(v, o, where)
    assert o.st.episode.max() >= 2


@pytest.mark.parametrize('cfg,n,steps', [('pogo10', 8192, 330), ('axe10', 4096, 250), ('bow20', 1024, 120), ('add32', 128, 70),
                                         ('fencer10m', 2048, 250), ('fire10h', 2048, 250), ('crate12h', 1024, 150)])
def test_fused_rollout_matches_oracle(cfg, n, steps):
    """ngw_rollout: T steps in one launch with in-kernel uniform actions == oracle stepping the same action stream."""
    spec = T.build_spec(cfg)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=5, autoreset=True, horizon=100, env_index_base=1 << 33)
    o = Oracle(spec.compile(), n, seed=5, autoreset=True, horizon=100, env_index_base=1 << 33)
    v.reset()
    o.reset()
    t0 = 0
    for chunk in (1, 2, steps - 3):
        v.rollout(chunk, action_seed=1234, t0=t0)
        assert o.rollout(chunk, 1234, t0) == 0
        t0 += chunk
        assert_state_equal(v, o, '%s rollout to t=%d' % (cfg, t0))
        reward, done, info = v.get_step_out()
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all()
        assert (info['message_code'] == o.msg_code).all() and (info['step_cost_code'] == o.cost_code).all()
    assert v.error_flags() == 0


@pytest.mark.parametrize('cfg,n,steps,horizon,every,depth', [('pogo10', 4096, 260, 50, 8, 1), ('pogo10', 3000, 120, 7, 1, 1), ('fire10h', 4096, 200, 50, 4, 1),
                                                             ('add32', 256, 60, 25, 16, 1), ('bow20', 1000, 120, 30, 64, 1), ('pogo13', 777, 100, 20, 3, 1),
                                                             ('fencer10m', 1500, 120, 30, 5, 1),
                                                             ('pogo10', 3000, 120, 7, 16, 4), ('fire10h', 4096, 200, 50, 16, 4), ('fire10h', 2000, 150, 30, 9, 2),
                                                             ('add32', 256, 60, 12, 30, 4), ('bow20', 1000, 120, 9, 20, 2), ('fencer10m', 1500, 120, 10, 25, 8),
                                                             ('axe10', 1000, 150, 11, 30, 4), ('pogo13', 777, 100, 6, 20, 4),
                                                             # rows of more than 64 sixteen-byte chunks: the wave copies them in several rounds
                                                             ('add36e', 200, 50, 12, 10, 2), ('add36e', 130, 40, 9, 8, 1)])
def test_prepared_next_episodes_are_bit_identical(cfg, n, steps, horizon, every, depth):
    """ngw_set_reset_prefetch (+ _depth: several episodes ahead per env): resets served from the shadow rows (staggered episode
    ends, instant deaths, several episode ends of one env between two refills, explicit masked resets, a stale row after
    set_state(episode=...), fused rollout, graph replay) give exactly the oracle's states."""
    import torch
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=21, autoreset=True, horizon=horizon, reset_prefetch=every, reset_prefetch_depth=depth)
    assert v.reset_prefetch_depth == depth
    o = Oracle(spec.compile(), n, seed=21, autoreset=True, horizon=horizon)
    v.reset(); o.reset()
    stag = (np.arange(n) * 7919 % horizon).astype(np.int32)            # episode ends spread over the batch
    v.set_state(0, step_count=stag); o.st.step_count[:] = stag
    rs = np.random.RandomState(4)
    for t in range(steps):
        if t == steps // 3:                                             # explicit masked reset in between
            mask = (np.arange(n) % 5 == 1).astype(np.uint8)
            v.reset(mask); o.reset(mask)
        if t == steps // 2:                                             # make every prepared row stale for a third of the envs
            ep = o.st.episode.copy(); ep[::3] += 1000
            v.set_state(0, episode=ep); o.st.episode[:] = ep
        a = rs.randint(0, A, size=n).astype(np.int32)
        _, reward, done, info = v.step(a)
        assert o.step(a) == 0
        where = '%s step %d' % (cfg, t)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), where
        assert (info['message_code'] == o.msg_code).all(), where
        if t % 20 == 19 or t == steps - 1:
            assert_state_equal(v, o, where)
    v.rollout(2 * horizon + 3, action_seed=77, t0=9); assert o.rollout(2 * horizon + 3, 77, 9) == 0
    assert_state_equal(v, o, cfg + ' rollout')
    acts = torch.randint(0, A, (6, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.graph_build(acts.data_ptr(), n, 6); v.graph_launch(horizon // 6 + 3)
    an = acts.cpu().numpy()
    for rep in range(horizon // 6 + 3):
        for t in range(6):
            o.step(an[t])
    assert_state_equal(v, o, cfg + ' graph')
    v.set_reset_prefetch(0)                                             # off again: inline resets
    for t in range(horizon + 2):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a)
    assert_state_equal(v, o, cfg + ' prefetch off')
    assert o.st.episode.min() >= 3 and v.error_flags() == 0


@pytest.mark.parametrize('cfg,n,horizon,prefetch', [('pogo10', 8192, 37, 0), ('axe10', 4096, 53, 16), ('bow20', 2048, 41, 0), ('fire10h', 4096, 29, 8),
                                                    ('crate10m', 4096, 61, 0), ('pogov0_10', 4096, 33, 4)])
def test_long_soak_matches_oracle(cfg, n, horizon, prefetch):
    """Soak: 4 000 batched steps (a hundred episodes per env, tens of millions of env-steps) in fused chunks of uneven
    length, then 300 per-launch steps; the full state equals the oracle's after every chunk."""
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=1234, autoreset=True, horizon=horizon, reset_prefetch=prefetch, env_index_base=7 * n)
    o = Oracle(spec.compile(), n, seed=1234, autoreset=True, horizon=horizon, env_index_base=7 * n)
    v.reset(); o.reset()
    t0 = 0
    for chunk in (1, 7, 100, 333, 1000, 2559):
        v.rollout(chunk, action_seed=99, t0=t0); assert o.rollout(chunk, 99, t0) == 0
        t0 += chunk
        assert_state_equal(v, o, '%s soak t=%d' % (cfg, t0))
    rs = np.random.RandomState(6)
    for t in range(300):
        a = rs.randint(0, A, size=n).astype(np.int32)
        _, reward, done, info = v.step(a); o.step(a)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), t
    assert_state_equal(v, o, cfg + ' soak end')
    assert o.st.episode.min() >= 4000 // horizon and v.error_flags() == 0


@pytest.mark.parametrize('cfg,n,T_,horizon', [('pogo10', 4096, 150, 40), ('bow20', 777, 60, 25), ('fire10h', 2048, 120, 30), ('add12m', 500, 40, 17)])
def test_rollout_with_supplied_actions_matches_oracle(cfg, n, T_, horizon):
    """ngw_rollout_actions: T steps in one launch taking the caller's [T, stride] int32 action rows == the oracle stepping
    through the same rows; a stride wider than n, an out-of-range action (env untouched for that step, flag raised)."""
    import time
    import torch
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=17, autoreset=True, horizon=horizon)
    o = Oracle(spec.compile(), n, seed=17, autoreset=True, horizon=horizon)
    v.reset(); o.reset()
    stride = n + 13
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    acts = torch.randint(0, A, (T_, stride), dtype=torch.int32, device='cuda', generator=g)
    torch.cuda.synchronize()
    v.rollout_actions(acts.data_ptr(), stride, 1)                       # a single step first, then the rest in one launch
    v.rollout_actions(acts[1:].data_ptr(), stride, T_ - 1)
    an = acts.cpu().numpy()
    for t in range(T_):
        assert o.step(np.ascontiguousarray(an[t, :n])) == 0
    assert_state_equal(v, o, cfg + ' rollout_actions')
    reward, done, info = v.get_step_out()
    assert (reward == o.reward).all() and (done == o.done.astype(bool)).all() and (info['message_code'] == o.msg_code).all()
    assert v.error_flags() == 0
    bad = acts[:3].clone(); bad[1, 5] = A                               # one invalid action in the middle row
    torch.cuda.synchronize()
    v.rollout_actions(bad.data_ptr(), stride, 3)
    bn = bad.cpu().numpy()
    for t in range(3):
        o.step(np.ascontiguousarray(bn[t, :n]))
    assert_state_equal(v, o, cfg + ' rollout_actions with an invalid action')
    assert v.error_flags() & 1
    with pytest.raises(ValueError):
        v.rollout_actions(acts.data_ptr(), n - 1, 2)


# ------------------------------------------------------------------ BASELINE sizes: size-independent properties
def test_full_size_properties_pogostick_65536():
    """BASELINE config 2 at full size: determinism, shard independence, structural invariants, oracle on a sample."""
    n = 65536
    spec = T.build_spec('pogo10')
    S, K = spec.map_size, len(spec.items_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100)
    v.reset()
    v.rollout(250, action_seed=1234)
    st = v.get_state()
    m = st['map'].reshape(n, S, S)
    wall = spec.items_id['wall']
    assert (m[:, 0, :] == wall).all() and (m[:, -1, :] == wall).all() and (m[:, :, 0] == wall).all() and (m[:, :, -1] == wall).all()
    assert (m[:, 1:-1, 1:-1] != wall).all() and (m >= 0).all() and (m < K).all()
    r, c = st['loc'][:, 0], st['loc'][:, 1]
    assert (m[np.arange(n), r, c] == 0).all()                      # the agent always stands on air
    assert (st['inv'] >= 0).all() and (st['step_count'] == 250 % 100).all() and (st['episode'] == 3).all()
    # conservation: tree_log on the map + in the inventory + consumed by Craft_plank never exceeds the 5 placed
    log = spec.items_id['tree_log']
    assert ((m == log).sum((1, 2)) + st['inv'][:, log] <= 5).all()
    # determinism + shard independence: two half-size handles keyed by global env index reproduce the halves
    for base in (0, n // 2):
        h = VecNovelGridworld(spec=spec, num_envs=n // 2, seed=0, autoreset=True, horizon=100, env_index_base=base)
        h.reset()
        h.rollout(250, action_seed=1234)
        hs = h.get_state()
        for k in STATE_KEYS:
            assert (hs[k] == st[k][base:base + n // 2]).all(), k
        h.close()
    # the oracle on a strided sample of envs (each env is independent, keyed by its global index)
    for e in (0, 1, 63, 64, 4097, 65535):
        o = Oracle(spec.compile(), 1, seed=0, autoreset=True, horizon=100, env_index_base=e)
        o.reset()
        o.rollout(250, 1234, 0)
        os_ = oracle_state(o)
        for k in STATE_KEYS:
            assert (os_[k][0] == st[k][e]).all(), (k, e)


@pytest.mark.parametrize('wl', ['C2', 'C3', 'C4', 'C5'])
def test_bench_call_sequence_matches_oracle_at_full_size(wl):
    """bench.py's own sequence at the workload's full size - reset, set_state(step_count) so that the horizon falls inside the timed
    launches, W = 5 then K = 20 one-step launches from ngw_step_device_many (eager, prepared rows consumed by the whole-wave copy at
    the horizon, default refill cadence), actions drawn on the device exactly as bench.py draws them - with EVERY env held to the
    oracle afterwards: state, episode counters and the last step's reward / done / info.  What the headline number times is what
    the parity suite checks."""
    import torch
    import bench
    from gym_novel_gridworlds_amd import apply_novelty
    env_id, S, nov, n, _ = bench.WORKLOADS[wl]
    spec = make_spec(env_id, S)
    if nov:
        np.random.seed(0)
        apply_novelty(spec, *nov)
    A = len(spec.actions_id)
    H, steps, warmup = bench.HORIZON, 20, 5
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=H)
    o = Oracle(spec.compile(), n, seed=0, autoreset=True, horizon=H)
    v.reset(); o.reset()
    hit = warmup + max(1, steps // 2)
    s0 = (H - hit) % H
    v.set_state(0, step_count=np.full(n, s0, np.int32))
    o.st.step_count[:] = s0
    g = torch.Generator(device='cuda')
    g.manual_seed(bench.ACTION_SEED)
    acts = torch.randint(0, A, (warmup + steps, n), dtype=torch.int32, device='cuda', generator=g)
    torch.cuda.synchronize()
    ep0 = int(v.get_state(0, 1)['episode'][0])
    v.step_device_many(acts[0].data_ptr(), n, warmup)
    v.step_device_many(acts[warmup].data_ptr(), n, steps)
    v.sync()
    assert v.error_flags() == 0
    an = acts.cpu().numpy()
    for t in range(warmup + steps):
        o.step(an[t])
    assert_state_equal(v, o, 'bench sequence ' + wl)
    out = v.device_outputs()
    assert (out['reward'].cpu().numpy() == o.reward).all() and (out['done'].cpu().numpy() == o.done).all()
    assert (out['info'].cpu().numpy().view(np.uint32) == o.info).all()
    assert int(v.get_state(0, 1)['episode'][0]) == ep0 + 1          # the reset of every env sat inside the 20 timed launches


@pytest.mark.parametrize('cfg,n,H,prefetch', [('pogo10', 3000, 9, 'auto'), ('fire10h', 2000, 50, 4), ('bow20', 700, 7, 0), ('axe10', 1500, 11, 3),
                                              ('add32', 200, 5, 'auto'), ('pogo13', 900, 64, 'auto'), ('fencer10m', 640, 8, 2)])
def test_terminal_observations_under_autoreset(cfg, n, H, prefetch):
    """ngw_set_terminal_capture: an env that ends an episode in a step (done, a FireWall death, or the horizon cut) keeps the state that
    episode ENDED in - what a second oracle, stepped from the same pre-step state WITHOUT autoreset, holds - while step() returns the new
    episode's first observation as before.  Staggered episode ends (a few lanes of a wave reset per step), whole-wave ends, prepared
    rows on and off, staged and in-place step kernels, and fused rollouts (rows captured from LDS inside the T-loop)."""
    spec = T.build_spec(cfg)
    A, S, K = len(spec.actions_id), spec.map_size, len(spec.items_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=3, autoreset=True, horizon=H, reset_prefetch=prefetch, terminal_capture=True)
    o = Oracle(spec.compile(), n, seed=3, autoreset=True, horizon=H)
    o2 = Oracle(spec.compile(), n, seed=3, autoreset=False, horizon=0)
    v.reset(); o.reset()
    sc = (np.arange(n) * 7 % H).astype(np.int32)
    sc[:128] = 0                                                      # the first two waves end their episodes together (whole-wave copy)
    v.set_state(0, step_count=sc)
    o.st.step_count[:] = sc
    rs = np.random.RandomState(5)
    ended = 0
    for t in range(3 * H + 5 if H < 30 else 70):
        for dst, src in zip(o2.st.arrays() + [o2.st.episode], o.st.arrays() + [o.st.episode]):
            dst[...] = src
        a = rs.randint(0, A, size=n).astype(np.int32)
        if t % 11 == 3:                                               # goal items now and then: `done` endings, not only horizon cuts
            inv = o.st.inv.copy()
            inv[rs.randint(0, n, size=n // 50), spec.items_id[spec.goal_item_to_craft]] = 1
            o.st.inv[...] = inv; o2.st.inv[...] = inv
            v.set_state(0, inv=inv)
        obs, reward, done, info = v.step(a)
        o.step(a); o2.step(a)
        assert (done == o.done.astype(bool)).all() and (reward == o.reward).all(), t
        assert (info['_final_observation'] == done).all()
        idx = np.nonzero(done)[0]
        ended += idx.size
        fo = info['final_observation']
        assert (fo['map'].reshape(n, -1)[idx] == o2.st.map[idx]).all(), (t, 'map')
        assert (fo['agent_location'][idx] == o2.st.loc[idx]).all() and (fo['agent_facing_id'][idx] == o2.st.facing[idx]).all(), (t, 'pose')
        assert (fo['inventory_items_quantity'][idx] == o2.st.inv[idx]).all(), (t, 'inventory')
        dev = v.terminal_observation(device=True)
        assert (dev['map'].cpu().numpy() == fo['map']).all() and (dev['inventory_items_quantity'].cpu().numpy() == fo['inventory_items_quantity']).all()
        assert (obs['map'].reshape(n, -1) == o.st.map).all() and (obs['agent_location'] == o.st.loc).all(), (t, 'the returned observation is the new episode')
    assert_state_equal(v, o, 'terminal capture ' + cfg)
    assert ended > n                                                  # every env ended at least one episode on average
    # fused rollouts capture too (the resetting lane stores its rows from LDS before the next episode overwrites them): the caller's action
    # rows, every step mirrored on the two oracles; a row keeps its value until its env ends an episode again
    import torch
    Tn = 2 * H + 3 if H < 30 else 40
    acts = torch.randint(0, A, (Tn, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    an = acts.cpu().numpy()
    exp = v.terminal_observation()
    v.rollout_actions(acts.data_ptr(), n, Tn)
    for t in range(Tn):
        for dst, src in zip(o2.st.arrays() + [o2.st.episode], o.st.arrays() + [o.st.episode]):
            dst[...] = src
        o.step(an[t]); o2.step(an[t])
        idx = np.nonzero(o.done)[0]
        exp['map'].reshape(n, -1)[idx] = o2.st.map[idx]
        exp['agent_location'][idx] = o2.st.loc[idx]; exp['agent_facing_id'][idx] = o2.st.facing[idx]
        exp['inventory_items_quantity'][idx] = o2.st.inv[idx]
    got = v.terminal_observation()
    for k in exp:
        assert (got[k] == exp[k]).all(), ('rollout', k)
    assert_state_equal(v, o, 'rollout with the capture on ' + cfg)
    v.rollout(H + 1, action_seed=9, t0=3); o.rollout(H + 1, 9, 3)      # (generated actions: every env ends an episode; the state stays exact)
    assert_state_equal(v, o, 'generated-action rollout with the capture on ' + cfg)
    v.set_terminal_capture(False)
    v.rollout(5, action_seed=1, t0=0); o.rollout(5, 1, 0)
    assert_state_equal(v, o, 'rollout after the capture was switched off')


def test_thirty_million_envs_match_oracle_slices():
    """33 554 432 envs on one GPU straight through the C-ABI (5.6 GB of state, 524 288 wavefronts; the map array alone
    exceeds 2^31 bytes - the 288 GB of HBM would hold 50x that): slices at the start, across the 2^31-byte mark of the map
    array, in the middle and at the very end equal the oracle run on those global env indices alone."""
    import ctypes as C
    from gym_novel_gridworlds_amd import _cabi
    n, base = 1 << 25, 1 << 40
    spec = T.build_spec('pogo10')
    cs = spec.compile()
    S2, K = spec.map_size ** 2, len(spec.items_id)
    L, h = _cabi.lib(), C.c_void_p()
    _cabi.check(L.ngw_create(C.byref(cs), n, 0, 5, base, C.byref(h)))
    try:
        _cabi.check(L.ngw_set_autoreset(h, 1, 13))
        _cabi.check(L.ngw_reset(h, None))
        _cabi.check(L.ngw_rollout(h, 30, 77, 3))
        flags = C.c_uint32(1)
        _cabi.check(L.ngw_error_flags(h, C.byref(flags)))
        assert flags.value == 0
        over = (1 << 31) // S2                                          # env whose map row straddles byte 2^31
        for first in (0, over - 1000, n // 2 + 17, n - 2048):
            cnt = 2048
            st = dict(map=np.zeros((cnt, S2), np.int8), loc=np.zeros((cnt, 2), np.int32), facing=np.zeros(cnt, np.int32),
                      inv=np.zeros((cnt, K), np.int32), selected=np.zeros(cnt, np.int32), step_count=np.zeros(cnt, np.int32),
                      episode=np.zeros(cnt, np.uint32))
            _cabi.check(L.ngw_get_state(h, first, cnt, *[st[k].ctypes.data for k in STATE_KEYS]))
            o = Oracle(cs, cnt, seed=5, autoreset=True, horizon=13, env_index_base=base + first)
            o.reset()
            assert o.rollout(30, 77, 3) == 0
            os_ = oracle_state(o)
            for k in STATE_KEYS:
                assert (os_[k].reshape(st[k].shape) == st[k]).all(), (k, first)
    finally:
        L.ngw_destroy(h)


def test_full_size_bow_65536_and_device_views():
    """BASELINE config 3 at full size + zero-copy device observation == host observation."""
    import torch
    n = 65536
    spec = T.build_spec('bow20')
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=11, autoreset=True, horizon=100)
    v.reset()
    acts = torch.randint(0, len(spec.actions_id), (30, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    for t in range(30):
        v.step_device(acts[t].data_ptr())
    v.sync()
    obs = v.get_observation(copy=True)
    dev = v.device_observation()
    assert (dev['map'].cpu().numpy() == obs['map']).all()
    assert (dev['agent_location'].cpu().numpy() == obs['agent_location']).all()
    assert (dev['agent_facing_id'].cpu().numpy() == obs['agent_facing_id']).all()
    assert (dev['inventory_items_quantity'].cpu().numpy() == obs['inventory_items_quantity']).all()
    o = Oracle(spec.compile(), 256, seed=11, autoreset=True, horizon=100)
    o.reset()
    a = acts[:, :256].cpu().numpy()
    for t in range(30):
        o.step(a[t])
    assert (o.st.map.reshape(256, 20, 20) == obs['map'][:256]).all() and (o.st.inv == obs['inventory_items_quantity'][:256]).all()


# ------------------------------------------------------------------ error behaviour
def test_invalid_action_raises_like_reference():
    ref = T.spec_json()['cfgs']['pogo10']
    v = VecNovelGridworld(num_envs=3, seed=1)
    v.reset()
    before = v.get_state()
    for a, exc, text in ref['invalid_action_errors']:
        with pytest.raises(ValueError, match=text):
            v.step(np.array([0, a, 1], np.int32))
    after = v.get_state()
    for k in STATE_KEYS:
        assert (before[k] == after[k]).all()


def test_invalid_device_action_sets_flag_and_leaves_env_untouched():
    import torch
    v = VecNovelGridworld(num_envs=130, seed=1)
    v.reset()
    before = v.get_state()
    a = torch.ones(130, dtype=torch.int32, device='cuda')
    a[77] = 17
    torch.cuda.synchronize()
    v.step_device(a.data_ptr())
    assert v.error_flags() == 1
    after = v.get_state()
    assert (after['facing'][77] == before['facing'][77]) and after['step_count'][77] == 0
    assert (np.delete(after['step_count'], 77) == 1).all()


def test_placement_exhaustion_raises_like_reference():
    with pytest.raises(AssertionError, match="Cannot place items, increase map size!"):
        VecNovelGridworld(num_envs=64, map_size=6).reset()
    # 16 candidates for 6 items: some envs run out of candidates, some do not.  The call raises (the sticky flag), and env by
    # env the outcome is the oracle's: the envs that could place everything hold exactly the oracle's episode, the others
    # stop where the reference's loop raised (pose drawn, the items placed so far on the map).
    from oracle.ngw_oracle import State, lib as olib
    import ctypes
    n, S = 200, 8
    spec = make_spec(T.POGO, S)
    cs = spec.compile()
    for fast in ('1', '2'):                                      # general kernel (the default at this size), dedicated kernel
        old = os.environ.get('NGW_FAST_RESET')
        os.environ['NGW_FAST_RESET'] = fast
        try:
            v = VecNovelGridworld(spec=spec, num_envs=n, seed=3)
        finally:
            if old is None:
                del os.environ['NGW_FAST_RESET']
            else:
                os.environ['NGW_FAST_RESET'] = old
        st = State(n, S, len(spec.items_id))
        failed = np.zeros(n, bool)
        for i in range(n):
            failed[i] = olib().ngwo_reset_philox(ctypes.byref(cs), 3, i, 1, st.map[i], st.loc[i], st.facing[i:i + 1], st.inv[i],
                                                 st.selected[i:i + 1], st.step_count[i:i + 1]) != 0
        assert failed.any() and not failed.all()
        with pytest.raises(AssertionError, match="Cannot place items, increase map size!"):
            v.reset()
        got = v.get_state()
        ok = ~failed
        assert (got['map'][ok] == st.map[ok]).all() and (got['loc'][ok] == st.loc[ok]).all() and (got['facing'][ok] == st.facing[ok]).all()
        assert (got['inv'][ok] == st.inv[ok]).all() and (got['episode'] == 1).all()
        assert (got['loc'][failed] == st.loc[failed]).all() and (got['facing'][failed] == st.facing[failed]).all()
        assert (got['map'][failed] == st.map[failed]).all(), 'fast=%s: a failed env holds other items than the reference had placed' % fast
        v.close()


def test_graph_stepping_equals_eager_stepping():
    """ngw_graph_build/launch replays exactly the step launches it captured (two replays == 2 x 6 eager steps)."""
    import torch
    spec = T.build_spec('axe10')
    n, A = 3000, len(spec.actions_id)
    acts = torch.randint(0, A, (6, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    a = VecNovelGridworld(spec=spec, num_envs=n, seed=21, autoreset=True, horizon=5)
    b = VecNovelGridworld(spec=spec, num_envs=n, seed=21, autoreset=True, horizon=5)
    a.reset()
    b.reset()
    a.graph_build(acts.data_ptr(), n, 6)
    a.graph_launch(2)
    for rep in range(2):
        for t in range(6):
            b.step_device(acts[t].data_ptr())
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert (sa[k] == sb[k]).all(), k
    ra, rb = a.get_step_out(copy=True), b.get_step_out(copy=True)
    assert (ra[0] == rb[0]).all() and (ra[1] == rb[1]).all()
    assert a.error_flags() == 0


@pytest.mark.parametrize('cfg', ['pogo10', 'bow20', 'axe10', 'add12m'])
def test_single_env_adapter_on_hip_backend(cfg):
    """BASELINE config 1 plumbing: the reference-shaped single-env API on the real device backend."""
    assert T.replay_adapter(cfg, 'hip', max_steps=250, n_single=150) > 300


@pytest.mark.parametrize('cfg,timeout_us', [('pogo10', ''), ('axe10', '20'), ('bow20', '5'), ('fire10h', '2000'), ('add12m', '0')])
def test_resident_step_loop_of_one_env_handles(cfg, timeout_us, monkeypatch):
    """The one-env handle's resident step loop (csrc/ngw_solo.inc): every step's outcome is looked up in the records the device speculated
    and the posted action committed behind the caller's back.  Held to the oracle step by step through what can go wrong around it: the
    loop ending on its idle limit at any point of the exchange (limits of 0 - 20 us make it end between, during and right after steps;
    the caller also pauses now and then), other calls on the handle in between (state reads and writes, resets, device-side launches),
    episodes that end (`done` is sticky under the reference semantics), invalid actions."""
    import time
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle
    if timeout_us:
        monkeypatch.setenv('NGW_SOLO_TIMEOUT_US', timeout_us)
    spec = T.build_spec(cfg)
    A, S, K = len(spec.actions_id), spec.map_size, len(spec.items_id)
    v = G.VecNovelGridworld(spec=spec, num_envs=1, seed=3)
    o = Oracle(spec.compile(), 1, seed=3)
    v.reset1(); o.reset()
    rs = np.random.RandomState(11)

    def same_state(where):
        mb, r, c, f, ib, sel, steps = v.last_state1()
        assert (np.frombuffer(mb, np.int8) == o.st.map[0]).all() and (r, c, f) == (o.st.loc[0][0], o.st.loc[0][1], o.st.facing[0]), where
        assert (np.frombuffer(ib, np.int32) == o.st.inv[0]).all() and sel == o.st.selected[0] and steps == o.st.step_count[0], where

    for t in range(1500):
        a = int(rs.randint(0, A))
        got = v.step1(a); o.step(np.array([a], np.int32))
        exp = (int(o.reward[0]), bool(o.done[0]), bool(o.result[0]), int(o.cost_code[0]), int(o.msg_code[0]), int(o.msg_arg[0]))
        assert got == exp, (t, a, got, exp)
        same_state(t)
        k = rs.randint(0, 60)
        if k == 0:
            time.sleep(0.002)                                           # well past any idle limit
        elif k == 1:                                                    # a state read of the device's copy: the loop is stopped first
            st = v.get_state()
            assert (st['map'] == o.st.map).all() and (st['inv'] == o.st.inv).all() and (st['step_count'] == o.st.step_count).all(), t
        elif k == 2:
            v.reset1(); o.reset()
        elif k == 3:                                                    # state injection between steps
            inv = rs.randint(0, 9, (1, K)).astype(np.int32)
            v.set_state(0, inv=inv); o.st.inv[:] = inv
        elif k == 4:                                                    # a per-launch device-side step on the same handle
            import torch
            b = rs.randint(0, A, 1).astype(np.int32)
            bd = torch.from_numpy(b).cuda()
            torch.cuda.synchronize()
            v.step_device(bd.data_ptr()); v.sync(); o.step(b)
        elif k == 5:
            with pytest.raises(ValueError, match='%d is not in list' % A):
                v.step1(A)
        if o.done[0] and k % 3 == 0:
            v.reset1(); o.reset()
    assert v.error_flags() == 0
    v.close()


def test_single_env_random_action_loop_shape():
    """tests/random_action.py:51-64 loop shape on the adapter: 50 steps, map_size change + reset every 10."""
    import gym_novel_gridworlds_amd as G
    np.random.seed(0)
    env = G.make('NovelGridworld-Pogostick-v1')
    obs = env.reset()
    for i in range(50):
        a = env.action_space.sample()
        obs, reward, done, info = env.step(a)
        assert set(obs) == {'map', 'agent_location', 'agent_facing_id', 'inventory_items_quantity'}
        assert obs['map'].shape == (env.map_size, env.map_size) and set(info) == {'result', 'step_cost', 'message'}
        if (i + 1) % 10 == 0:
            env.map_size = int(np.random.randint(low=10, high=20, size=1)[0])
            obs = env.reset()
            wall = env.items_id['wall']
            assert (obs['map'][0] == wall).all() and (obs['map'] == env.items_id['tree_log']).sum() == 5
    env.close()


def test_packed_observation_gather_roundtrip_single_rank():
    """dist.ShardedVecNovelGridworld on one GPU: packed observation (the RCCL gather payload) unpacks to the batch."""
    from gym_novel_gridworlds_amd.dist import ShardedVecNovelGridworld
    spec = T.build_spec('axe10')
    env = ShardedVecNovelGridworld(global_num_envs=2048, spec=spec, seed=9, autoreset=True, horizon=20)
    env.reset()
    env.rollout(33, action_seed=7)
    got = env.gather_observation()
    st = env.local.get_state()
    reward, done, info = env.local.get_step_out(copy=True)
    assert (got['map'].cpu().numpy().reshape(2048, -1) == st['map']).all()
    assert (got['agent_location'].cpu().numpy() == st['loc']).all() and (got['agent_facing_id'].cpu().numpy() == st['facing']).all()
    assert (got['inventory_items_quantity'].cpu().numpy() == st['inv']).all()
    assert (got['reward'].cpu().numpy() == reward).all() and (got['done'].cpu().numpy() == done).all()
    env.close()


def test_full_size_axe_medium_32768_per_gpu():
    """BASELINE config 4 at its per-GPU size (262 144 envs over 8 GPUs = 32 768 each): the rank-3 shard of the global
    batch, keyed by global env index, against the oracle on a sample + entity pick-up invariants."""
    n, base = 32768, 3 * 32768
    spec = T.build_spec('axe10')
    axe = spec.items_id['wooden_axe']
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=100, env_index_base=base)
    v.reset()
    st0 = v.get_state()
    assert ((st0['map'] == axe).sum(1) == 1).all()                 # exactly one axe lies on every fresh map
    v.rollout(160, action_seed=1234)
    st = v.get_state()
    on_map, held = (st['map'] == axe).sum(1), st['inv'][:, axe]
    assert ((on_map + held) == 1).all()                            # the axe is either on the map or in the inventory
    assert (st['step_count'] == 60).all() and (st['episode'] == 2).all()
    for e in (0, 17, 4095, 32767):
        o = Oracle(spec.compile(), 1, seed=0, autoreset=True, horizon=100, env_index_base=base + e)
        o.reset()
        o.rollout(160, 1234, 0)
        os_ = oracle_state(o)
        for k in STATE_KEYS:
            assert (os_[k][0] == st[k][e]).all(), (k, e)


def test_full_size_additem_hard_32x32_65536_per_gpu():
    """BASELINE config 5 at its per-GPU size (65 536 envs, 32x32): AddItem coverage statistics + oracle sample."""
    n = 65536
    spec = T.build_spec('add32')
    arrow, wall = spec.items_id['arrow'], spec.items_id['wall']
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=4, autoreset=True, horizon=100)
    v.reset()
    st = v.get_state()
    m = st['map'].reshape(n, 32, 32)
    assert (m[:, 0, :] == wall).all() and (m[:, :, -1] == wall).all()
    cnt = (m == arrow).sum((1, 2))
    # 894 air cells, pct in [20, 30): ceil(894 * pct / 100) in [179, 260], minus at most the agent cell
    assert cnt.min() >= 178 and cnt.max() <= 260
    assert (m[np.arange(n), st['loc'][:, 0], st['loc'][:, 1]] == 0).all()
    full = np.array([int(np.ceil(894 * (pct / 100))) for pct in range(20, 30)])      # items placed before the agent-cell skip
    which = np.searchsorted(full, cnt)                             # cnt is full[p] or full[p] - 1 (agent cell among the chosen)
    assert ((full[which] == cnt) | (full[which] - 1 == cnt)).all()
    hist = np.bincount(which, minlength=10)
    assert hist.min() > 0.08 * n and hist.max() < 0.12 * n         # randint(20, 30): ten equally likely percentages
    v.rollout(40, action_seed=99)
    st = v.get_state()
    for e in (0, 63, 64, 12345, 65535):
        o = Oracle(spec.compile(), 1, seed=4, autoreset=True, horizon=100, env_index_base=e)
        o.reset()
        o.rollout(40, 99, 0)
        os_ = oracle_state(o)
        for k in STATE_KEYS:
            assert (os_[k][0] == st[k][e]).all(), (k, e)


@pytest.mark.parametrize('env_id,S', [(T.POGO, 40), (T.BOW, 45), (T.POGO, 9), (T.BOW, 8)])
def test_extreme_map_sizes_match_oracle(env_id, S):
    """Largest maps the one-wave LDS budget admits (and the smallest that can hold the items): reset + steps vs oracle."""
    spec = make_spec(env_id, S)
    n = 200
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=2, autoreset=True, horizon=30)
    o = Oracle(spec.compile(), n, seed=2, autoreset=True, horizon=30)
    ok = o.reset() == 0
    try:
        v.reset()
        assert ok
    except AssertionError:
        assert not ok
        return
    v.rollout(75, action_seed=5)
    oflags = o.rollout(75, 5, 0)
    # tiny maps may exhaust placement on a later (auto)reset: both sides must then raise the same sticky flag, and the state is
    # compared either way (a failed placement leaves the same partial map on both sides)
    assert v.error_flags() == oflags
    assert_state_equal(v, o, '%s S=%d' % (env_id, S))


def test_maps_beyond_the_lds_budget_step_and_reset_but_refuse_fused_rollouts():
    """Beyond ~46 x 46 a wave's 64 maps no longer fit in LDS.  The no-stage step kernel and the dedicated new-episode kernel do
    not keep them there: such a handle resets and steps (equal to the oracle, autoreset and prepared episodes included) and
    refuses the calls that would need the maps in LDS; configurations whose resets need the general kernel are refused at creation."""
    if os.environ.get('NGW_FAST_RESET') == '0' or os.environ.get('NGW_NOSTAGE') == '0':
        pytest.skip('the A/B switches that route everything through kernels with the maps in LDS keep the 160 KiB limit')
    from gym_novel_gridworlds_amd import apply_novelty
    for S, nov in ((60, None), (52, ('additem', 'easy', 'arrow', '')), (64, ('firewall', 'medium', '', ''))):
        spec = make_spec(T.POGO, S)
        if nov:
            apply_novelty(spec, *nov)
        A, n = len(spec.actions_id), 300
        v = VecNovelGridworld(spec=spec, num_envs=n, seed=9, autoreset=True, horizon=20, reset_prefetch=7)
        o = Oracle(spec.compile(), n, seed=9, autoreset=True, horizon=20)
        v.reset(); assert o.reset() == 0
        assert_state_equal(v, o, 'S=%d reset' % S)
        rs = np.random.RandomState(3)
        for t in range(50):
            a = rs.randint(0, A, size=n).astype(np.int32)
            _, reward, done, info = v.step(a); o.step(a)
            assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), (S, t)
        assert_state_equal(v, o, 'S=%d stepped' % S)
        with pytest.raises(ValueError, match='LDS'):
            v.rollout(5)
        v.close()
    spec = make_spec(T.POGO, 60)
    apply_novelty(spec, 'fence', 'easy', 'oak', '')                 # a pass that reads the map: general kernel only
    with pytest.raises(ValueError, match='LDS'):
        VecNovelGridworld(spec=spec, num_envs=64)


@pytest.mark.parametrize('cfg,n,T_,horizon,prefetch,supplied', [('pogo10', 2048, 150, 40, 0, False), ('axe10', 1000, 230, 100, 'auto', False),
                                                               ('bow20', 512, 90, 30, 16, True), ('fire10h', 1024, 100, 25, 0, True)])
def test_rollout_rows_and_episode_accumulators_match_oracle(cfg, n, T_, horizon, prefetch, supplied):
    """ngw_rollout_outputs: every step of a fused rollout leaves its reward / done row, and the per-env accumulators (running
    return / length, sum of returns / count of finished episodes) equal the oracle stepped one step at a time."""
    import torch
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=11, autoreset=True, horizon=horizon, reset_prefetch=prefetch)
    o = Oracle(spec.compile(), n, seed=11, autoreset=True, horizon=horizon)
    v.reset(); o.reset()
    stride = n + 7
    rew = torch.full((T_, stride), -99, dtype=torch.int32, device='cuda')
    dn = torch.full((T_, stride), 9, dtype=torch.uint8, device='cuda')
    acts = torch.randint(0, A, (T_, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.rollout_outputs(rew.data_ptr(), dn.data_ptr(), stride, accumulate=True)
    want_r, want_d = np.zeros((T_, n), np.int32), np.zeros((T_, n), np.uint8)
    run_ret, run_len, sum_ret, n_eps = (np.zeros(n, np.int64) for _ in range(4))
    an = acts.cpu().numpy()
    t = 0
    for chunk in (1, 3, T_ - 4):                                        # the accumulators carry over rollout calls; rows restart at 0
        if supplied:
            v.rollout_actions(acts[t].data_ptr(), n, chunk)
        else:
            v.rollout(chunk, action_seed=5, t0=t)
        for k in range(chunk):
            if supplied:
                o.step(an[t + k])
            else:
                o.rollout(1, 5, t + k)
            want_r[t + k], want_d[t + k] = o.reward, o.done
            run_ret += o.reward; run_len += 1
            fin = o.done.astype(bool)
            sum_ret[fin] += run_ret[fin]; n_eps[fin] += 1; run_ret[fin] = 0; run_len[fin] = 0
        v.sync()
        got_r, got_d = rew.cpu().numpy(), dn.cpu().numpy()
        assert (got_r[:chunk, :n] == want_r[t:t + chunk]).all() and (got_d[:chunk, :n] == want_d[t:t + chunk]).all(), (cfg, t)
        assert (got_r[:, n:] == -99).all() and (got_d[:, n:] == 9).all()             # nothing outside the rows' env columns
        t += chunk
    st = v.episode_stats(clear=True)
    assert (st['run_return'] == run_ret).all() and (st['run_length'] == run_len).all()
    assert (st['sum_return'] == sum_ret).all() and (st['n_episodes'] == n_eps).all() and n_eps.max() >= 2
    assert all((x == 0).all() for x in v.episode_stats().values())
    v.rollout_outputs()                                                  # off again: rollouts leave the rows alone
    rew.fill_(-7); torch.cuda.synchronize()
    v.rollout(5, action_seed=5, t0=t); o.rollout(5, 5, t)
    v.sync()
    assert (rew.cpu().numpy() == -7).all()
    assert_state_equal(v, o, cfg + ' after the rows')


def test_autoreset_switches_prepared_episodes_on_at_the_c_abi():
    """ngw_set_autoreset alone (no ngw_set_reset_prefetch) prepares next episodes: staggered episode ends are served from the
    shadow rows - visible as the refill launches' effect on the shadow tags - and a caller's own cadence (incl. 0) is kept."""
    import ctypes as C
    from gym_novel_gridworlds_amd import _cabi
    v = VecNovelGridworld(num_envs=256, seed=1)                          # autoreset off: nothing prepared
    L = _cabi.lib()
    assert v.reset_prefetch == 0
    _cabi.check(L.ngw_set_autoreset(v._h, 1, 100))
    o = Oracle(v.spec.compile(), 256, seed=1, autoreset=True, horizon=100)
    v.reset(); o.reset()
    v.rollout(230, action_seed=3); o.rollout(230, 3, 0)
    assert_state_equal(v, o, 'default cadence')
    v.set_reset_prefetch(0)                                             # the caller's choice survives another ngw_set_autoreset
    _cabi.check(L.ngw_set_autoreset(v._h, 1, 80))
    o.horizon = 80
    v.rollout(200, action_seed=3, t0=230); o.rollout(200, 3, 230)
    assert_state_equal(v, o, 'caller cadence kept')
    assert v.error_flags() == 0


@pytest.mark.parametrize('cfg,n,prefetch', [('pogo10', 50, 4), ('axe10', 64, 0), ('bow20', 33, 8), ('fire10h', 40, 0),
                                            ('pogo10', 1, 4), ('add32', 1, 2), ('pogo13', 1, 0), ('add36e', 3, 3), ('bow20', 1, 1 << 20)])
def test_one_wavefront_handles_with_host_mirror_match_oracle(cfg, n, prefetch):
    """Handles of at most 64 envs (the gym.Env adapter's fast path: n = 1, whose action travels in the kernel arguments) keep a
    mirror of their state in GPU-addressable host memory that every host step / reset refreshes: the observation a step returns
    (read from the mirror), steps with autoreset, prepared episodes, a fused rollout, masked resets and state round trips equal
    the oracle."""
    spec = T.build_spec(cfg)
    A, S = len(spec.actions_id), spec.map_size
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=9, autoreset=True, horizon=10, reset_prefetch=prefetch)
    o = Oracle(spec.compile(), n, seed=9, autoreset=True, horizon=10)

    def check_obs(obs, where):
        assert (obs['map'].reshape(n, -1) == o.st.map).all() and (obs['agent_location'] == o.st.loc).all(), where
        assert (obs['agent_facing_id'] == o.st.facing).all() and (obs['inventory_items_quantity'] == o.st.inv).all(), where

    obs = v.reset(); o.reset()
    check_obs(obs, 'reset')
    rs = np.random.RandomState(2)
    for t in range(60):
        a = rs.randint(0, A, size=n).astype(np.int32)
        obs, reward, done, info = v.step(a); o.step(a)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all() and (info['message_code'] == o.msg_code).all(), t
        check_obs(obs, 'step %d' % t)
        if t == 30:
            mask = (np.arange(n) % 3 == 0).astype(np.uint8)
            obs = v.reset(mask); o.reset(mask)
            check_obs(obs, 'masked reset')
    assert_state_equal(v, o, cfg + ' steps')
    v.rollout(37, action_seed=4, t0=3); o.rollout(37, 4, 3)
    assert_state_equal(v, o, cfg + ' rollout')
    a = rs.randint(0, A, size=n).astype(np.int32)                      # the mirror follows again after launches that do not refresh it
    obs, _, _, _ = v.step(a); o.step(a)
    check_obs(obs, 'step after a rollout')
    st = v.get_state()
    v.set_state(0, **{k: st[k] for k in ('map', 'loc', 'facing', 'inv', 'selected', 'step_count', 'episode')})
    assert_state_equal(v, o, cfg + ' state round trip')
    if n == 1:                                                          # the adapter's calls
        for t in range(25):
            act = int(rs.randint(0, A))
            out = v.step1(act); o.step(np.array([act], np.int32))
            assert out[0] == int(o.reward[0]) and out[1] == bool(o.done[0]) and out[4] == int(o.msg_code[0]), t
            mb, r, c, f, ib, sel, steps = v.last_state1()
            assert mb == o.st.map[0].astype(np.int8).tobytes() and (r, c, f) == (int(o.st.loc[0][0]), int(o.st.loc[0][1]), int(o.st.facing[0])), t
            assert ib == o.st.inv[0].astype(np.int32).tobytes() and sel == int(o.st.selected[0]) and steps == int(o.st.step_count[0]), t
        v.reset1(); o.reset()
        assert v.last_state1()[0] == o.st.map[0].astype(np.int8).tobytes()
    with pytest.raises(ValueError):
        v.step(np.full(n, A, np.int32))                                 # invalid id: raised on the host, nothing stepped
    assert v.error_flags() == 0


def test_big_batch_host_step_uses_one_block_and_matches_oracle():
    """VecNovelGridworld.step() at 40 000 envs: the host arrays are the sections of one page-locked block (ngw_host_step_layout),
    the outputs come back with one pack launch + one copy; values equal the oracle's, also without the map (with_obs=False)."""
    n = 40000
    spec = T.build_spec('pogo10')
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=6, autoreset=True, horizon=9)
    o = Oracle(spec.compile(), n, seed=6, autoreset=True, horizon=9)
    v.reset(); o.reset()
    rs = np.random.RandomState(8)
    for t in range(25):
        a = rs.randint(0, A, size=n).astype(np.int32)
        obs, reward, done, info = v.step(a, with_obs=(t % 5 != 4)); o.step(a)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), t
        assert (info['result'] == o.result.astype(bool)).all() and (info['step_cost_code'] == o.cost_code).all() and (info['message_arg'] == o.msg_arg).all(), t
        if obs is not None:
            assert (obs['map'].reshape(n, -1) == o.st.map).all() and (obs['agent_location'] == o.st.loc).all()
            assert (obs['agent_facing_id'] == o.st.facing).all() and (obs['inventory_items_quantity'] == o.st.inv).all()
            ls = v.last_state()
            assert (ls['selected'] == o.st.selected).all() and (ls['step_count'] == o.st.step_count).all()
    assert_state_equal(v, o, 'one-block host steps')


@pytest.mark.parametrize('cfg,n,horizon,slices', [('pogo10', 40000, 9, ''), ('axe10', 20000, 40, ''), ('add32', 4096, 15, '4'), ('pogo13', 30000, 11, '3'), ('crate10m', 20000, 25, '1'),
                                                  ('fire10h', 9000, 30, '2'), ('bow20', 70000, 13, '4'), ('pogo13', 30000, 11, ''), ('add32', 4100, 15, ''), ('crate10m', 20000, 25, ''),
                                                  ('fire10h', 9000, 30, ''), ('bow20', 70000, 13, ''), ('fencer12h', 8000, 12, ''), ('pogo10', 40000, 9, 'delta'), ('add32', 4096, 15, 'delta'),
                                                  ('axe10', 20001, 14, 'inline'), ('add29h', 5000, 10, 'inline'), ('axe10', 20000, 21, 'terminal'), ('axe10', 12000, 0, 'noauto'), ('fire10h', 12000, 0, 'noauto')])
def test_host_step_delta_refresh_matches_oracle(cfg, n, horizon, slices, monkeypatch):
    """ngw_step_host on a big batch moves only what changed (include/ngw.h ngw_host_step_layout): the host observation equals
    the oracle's after EVERY step - through in-step resets, entity pick-ups and crates, odd row sizes - and after everything
    that invalidates the mirror in between (explicit resets, device steps, a fused rollout, state injection, refresh_host).
    `slices`: '' = the library's default, the step kernel's HOST WRITE-THROUGH form (one launch: the kernel stores what the step changes
    into the caller's block itself and its last block publishes a sequence number the call polls); 'delta' = NGW_HOST_WRITE_THROUGH=0, the delta
    kernel behind every step; a number = the pipelined form of that (the batch steps in slices while a second stream brings finished slices
    across PCIe; a slice is a multiple of 64 envs, so batch sizes that are no multiple of 64 x slices end in a short last slice); 'inline' =
    write-through without prepared episodes (the cold tail places the new episode inline); 'terminal' = with terminal-observation capture;
    'noauto' = the reference's semantics (no autoreset: `done` is sticky until the explicit resets in between)."""
    import torch
    kw = {}
    if slices == 'delta':
        monkeypatch.setenv('NGW_HOST_WRITE_THROUGH', '0')
    elif slices == 'inline':
        kw['reset_prefetch'] = 0
    elif slices == 'terminal':
        kw['terminal_capture'] = True
    elif slices:
        monkeypatch.setenv('NGW_API_SLICES', slices)
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    auto = slices != 'noauto'
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=16, autoreset=auto, horizon=horizon, **kw)
    o = Oracle(spec.compile(), n, seed=16, autoreset=auto, horizon=horizon)
    v.reset(); o.reset()
    if auto:
        stag = (np.arange(n) * 7919 % horizon).astype(np.int32)
        v.set_state(0, step_count=stag); o.st.step_count[:] = stag
    rs = np.random.RandomState(2)

    def host_equals_oracle(obs, where):
        assert (obs['map'].reshape(n, -1) == o.st.map).all() and (obs['agent_location'] == o.st.loc).all(), where
        assert (obs['agent_facing_id'] == o.st.facing).all() and (obs['inventory_items_quantity'] == o.st.inv).all(), where
        ls = v.last_state()
        assert (ls['selected'] == o.st.selected).all() and (ls['step_count'] == o.st.step_count).all(), where

    for t in range(60):
        if t == 20:
            mask = (np.arange(n) % 7 == 3).astype(np.uint8)
            v.reset(mask); o.reset(mask)
        if t == 30:
            acts = torch.randint(0, A, (3, n), dtype=torch.int32, device='cuda')
            torch.cuda.synchronize()
            v.step_device_many(acts.data_ptr(), n, 3)
            for i in range(3):
                o.step(acts[i].cpu().numpy())
        if t == 40:
            v.rollout(7, action_seed=5, t0=0); o.rollout(7, 5, 0)
        if t == 45:
            inv = o.st.inv.copy(); inv[::4, 2] += 3
            v.set_state(0, inv=inv); o.st.inv[:] = inv
        if t == 50:
            v._obs['map'][...] = 0                                      # the caller scribbles over the block ...
            v.refresh_host()                                            # ... and says so
        a = rs.randint(0, A, size=n).astype(np.int32)
        obs, reward, done, info = v.step(a); o.step(a)
        assert reward.dtype == np.int32                                 # (whatever the batch size)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), t
        assert (info['message_code'] == o.msg_code).all() and (info['message_arg'] == o.msg_arg).all(), t
        host_equals_oracle(obs, '%s step %d' % (cfg, t))
    assert (not auto or o.st.episode.min() >= 2) and v.error_flags() == 0
    assert_state_equal(v, o, cfg + ' delta host steps')


def test_refill_cadence_adapts_to_short_episodes_and_results_stay_exact():
    """Default prepared-episode setting: under FireWall an env ends several episodes between two refills, the stale rows are
    counted on the device and the host first keeps more episodes prepared per env (2, then 4), then shortens the refill
    cadence - without changing a single result."""
    if os.environ.get('NGW_ADAPT_PREFETCH') == '0':
        pytest.skip('adaptation is switched off (NGW_ADAPT_PREFETCH=0)')
    import ctypes
    from gym_novel_gridworlds_amd import _cabi
    spec = T.build_spec('fire10h')
    A = len(spec.actions_id)
    n = 2048
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=21, autoreset=True, horizon=100)
    o = Oracle(spec.compile(), n, seed=21, autoreset=True, horizon=100)
    lib = _cabi.lib()
    lib.ngw_debug_refill_cadence.argtypes = [ctypes.c_void_p]
    lib.ngw_debug_refill_cadence.restype = ctypes.c_int
    start = lib.ngw_debug_refill_cadence(v._h)
    assert start == v.reset_prefetch == 75               # the default under a horizon of 100: 3/4 of it
    v.reset(); o.reset()
    rs = np.random.RandomState(4)
    for t in range(400):
        a = rs.randint(0, A, size=n).astype(np.int32)
        _, reward, done, info = v.step(a); o.step(a)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), t
    assert_state_equal(v, o, 'fire10h adaptive cadence')
    # the host first prepares more episodes ahead per env (depth 1 -> 2 -> 4), then refills more often
    assert v.reset_prefetch_depth > 1 or lib.ngw_debug_refill_cadence(v._h) < start
    for t in range(600):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a)
    assert_state_equal(v, o, 'fire10h adaptive depth and cadence')
    assert v.reset_prefetch_depth == 4 and lib.ngw_debug_refill_cadence(v._h) <= start   # (2 048 envs: four rows ahead are enough at 75 steps)
    v.rollout(300, action_seed=8, t0=0); assert o.rollout(300, 8, 0) == 0
    assert_state_equal(v, o, 'fire10h adaptive cadence, fused')
    assert v.error_flags() == 0


@pytest.mark.parametrize('cfg,n,K', [('pogo10', 3000, 20), ('axe10', 2000, 7), ('bow20', 1500, 37)])
def test_short_graphs_keep_the_refill_cadence_between_replays(cfg, n, K):
    """A graph of at most half the refill cadence is captured without a refill of its own (ngw_graph_build: an "open" graph); ngw_graph_launch
    counts its steps and issues the refill between replays.  Same results as the oracle over many replays with episode ends spread over the
    batch, the prepared episodes keep up (no reset had to run its placement inline), and eager steps / resets mix in between."""
    import torch
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=21, autoreset=True, horizon=100)
    o = Oracle(spec.compile(), n, seed=21, autoreset=True, horizon=100)
    v.reset(); o.reset()
    stag = (np.arange(n) * 37 % 100).astype(np.int32)
    v.set_state(0, step_count=stag); o.st.step_count[:] = stag
    acts = torch.randint(0, A, (K, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    an = acts.cpu().numpy()
    for t in range(80):                                                 # (the stale rows of the state injection above are refilled on the way)
        a = an[t % K]
        v.step_device(acts[t % K].data_ptr()); o.step(a)
    v.graph_build(acts.data_ptr(), n, K)
    from gym_novel_gridworlds_amd import _cabi
    import ctypes
    lib = _cabi.lib()
    lib.ngw_debug_slow_resets.argtypes = [ctypes.c_void_p]
    lib.ngw_debug_slow_resets.restype = ctypes.c_longlong
    slow0 = lib.ngw_debug_slow_resets(v._h)
    for rep in range(30):
        v.graph_launch(1)
        for t in range(K):
            o.step(an[t])
        if rep % 7 == 3:
            v.step_device(acts[0].data_ptr()); o.step(an[0])
        if rep == 17:
            mask = (np.arange(n) % 5 == 2).astype(np.uint8)
            v.reset(mask); o.reset(mask)
        if rep % 5 == 0:
            assert_state_equal(v, o, '%s replay %d' % (cfg, rep))
    assert_state_equal(v, o, cfg + ' open graph replays')
    assert o.st.episode.min() >= 2 and v.error_flags() == 0
    assert 0 <= lib.ngw_debug_slow_resets(v._h) - slow0 <= n // 20     # (early `done`s and the masked reset's stale rows aside, every reset found its row)
    v.close()


def test_graph_replay_reports_do_not_lengthen_the_refill_cadence():
    """A replayed graph reports a whole replay's refills at once.  Averaged over that many, a cadence looks quiet that is noisy refill by
    refill: the host used to probe a longer cadence right after the capture, re-capture the graph at every change and take the change back
    from the eager steps around the replays (tools/x1_probe.py: 18 -> 36 -> 72 -> 36, one region at twice the time).  Such a report may
    tighten the cadence, never lengthen it - and whatever the cadence is, the state stays the oracle's."""
    if os.environ.get('NGW_ADAPT_PREFETCH') == '0':
        pytest.skip('adaptation is switched off (NGW_ADAPT_PREFETCH=0)')
    import torch
    spec = T.build_spec('fire10h')
    A = len(spec.actions_id)
    n, G = 65536, 300
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=33, autoreset=True, horizon=100)
    o = Oracle(spec.compile(), n, seed=33, autoreset=True, horizon=100)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    acts = torch.randint(0, A, (G, n), dtype=torch.int32, device='cuda', generator=g)
    torch.cuda.synchronize()
    an = acts.cpu().numpy()
    v.reset(); o.reset()
    for i in range(21):                                    # a live loop: a report per refill, the cadence tightens
        v.step_device_many(acts[100 * (i % 3)].data_ptr(), n, 100); v.sync()
        for t in range(100):
            o.step(an[100 * (i % 3) + t])
    before = v.refill_cadence
    assert before < v.reset_prefetch, 'FireWall hard at 65 536 envs tightens the refill cadence (precondition of this test)'
    v.graph_build(acts.data_ptr(), n, G)
    for rep in range(6):
        v.graph_launch(1); v.sync()
        for t in range(G):
            o.step(an[t])
        now = v.refill_cadence
        assert now <= before, 'replay %d lengthened the refill cadence: %d -> %d' % (rep, before, now)
        before = now
    assert_state_equal(v, o, 'fire10h, graph replays after eager adaptation')
    assert v.error_flags() == 0


@pytest.mark.parametrize('pack', ['1', '0'])
@pytest.mark.parametrize('cfg,n', [('add12m', 700), ('add18h', 500), ('add24m', 300), ('add29h', 200), ('add32', 260), ('add36e', 130), ('crate20h', 400),
                                   ('add11e', 300), ('fire10h', 700), ('fire14m', 300), ('replwall12e', 300), ('fire32m', 130)])
def test_additem_new_episode_kernel_every_variant(cfg, n, pack, monkeypatch):
    """The subset pass of AddItem / Crate (air of the interior) and of FireWall / ReplaceItem of the wall ring, drawn without an
    index array (include/ngw.h, ngw_spec.n_passes): the dedicated new-episode kernel (one bit per cell; `pack` = '1') and the
    general kernel's byte-map form (NGW_FAST_RESET=0; also what a reset inside a step runs when no prepared episode exists)
    at every mask width (register masks of 2 / 8 words, LDS masks) and row alignment (16-byte pieces, dwords, bytes):
    explicit resets, masked resets, prepared episodes under autoreset and a fused rollout equal the oracle's Philox mode."""
    monkeypatch.setenv('NGW_FAST_RESET', pack)
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    # prepared next episodes on (a refill after every reset and every 8 steps): the second explicit reset below COPIES its rows
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=5, autoreset=True, horizon=12, env_index_base=3, reset_prefetch=8)
    o = Oracle(spec.compile(), n, seed=5, autoreset=True, horizon=12, env_index_base=3)
    for ep in range(2):
        v.reset(); assert o.reset() == 0
        assert_state_equal(v, o, '%s pack=%s reset %d' % (cfg, pack, ep))
    mask = (np.arange(n) % 5 == 1).astype(np.uint8)
    v.reset(mask); o.reset(mask)
    assert_state_equal(v, o, '%s pack=%s masked reset' % (cfg, pack))
    rs = np.random.RandomState(8)
    for t in range(40):
        a = rs.randint(0, A, size=n).astype(np.int32)
        _, reward, done, info = v.step(a); o.step(a)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), t
    assert_state_equal(v, o, '%s pack=%s stepped' % (cfg, pack))
    v.reset(); assert o.reset() == 0                                   # rows consumed since the last refill are stale: copy and build mixed in a wave
    assert_state_equal(v, o, '%s pack=%s reset with some rows stale' % (cfg, pack))
    v.rollout(30, action_seed=3, t0=0); assert o.rollout(30, 3, 0) == 0
    assert_state_equal(v, o, '%s pack=%s fused' % (cfg, pack))
    assert v.error_flags() == 0


def test_staggered_episode_ends_are_served_by_prepared_episodes():
    """Episode ends spread over the batch (a learning loop): every reset must find its prepared row - the device counter of
    resets that ran the placement loop inside a step stays at zero - and the states equal the oracle's."""
    import ctypes
    from gym_novel_gridworlds_amd import _cabi
    lib = _cabi.lib()
    lib.ngw_debug_slow_resets.argtypes = [ctypes.c_void_p]
    lib.ngw_debug_slow_resets.restype = ctypes.c_longlong
    spec = T.build_spec('pogo10')
    A = len(spec.actions_id)
    n, H = 4096, 80                                              # (prepared episodes are the default from a horizon of 64)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=3, autoreset=True, horizon=H)
    o = Oracle(spec.compile(), n, seed=3, autoreset=True, horizon=H)
    v.reset(); o.reset()
    sc = (np.arange(n) * 7919 % H).astype(np.int32)
    v.set_state(0, step_count=sc); o.st.step_count[:] = sc
    s0 = lib.ngw_debug_slow_resets(v._h)                          # (explicit resets are not counted)
    assert s0 >= 0
    rs = np.random.RandomState(12)
    for t in range(3 * H):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a)
    assert_state_equal(v, o, 'staggered ends')
    assert o.st.episode.min() >= 3
    assert 0 <= lib.ngw_debug_slow_resets(v._h) - s0 <= n // 100  # (an early `done` inside one cadence is the only way to miss)


@pytest.mark.parametrize('cfg', T.G6_CFGS)
def test_gpu_resets_follow_the_reference_distribution(cfg):
    """G6: per-cell frequencies of the pass item and of the agent cell, and the histogram of the item
    count, of 65 536 GPU resets against 10 000 resets of the reference itself (tests/golden/g6_<cfg>.npz)."""
    import os
    spec = T.build_spec(cfg)
    S = spec.map_size
    n = 65536
    g = dict(np.load(os.path.join(T.GOLDEN, 'g6_%s.npz' % cfg)))
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=4242)
    v.reset()
    st = v.get_state()
    assert v.error_flags() == 0
    freq, hist, agent = T.g6_stats(st['map'], st['loc'], int(g['item']), S)
    mean_ref, mean_got = T.g6_check(cfg, freq, hist, agent, n)
    assert abs(mean_ref - mean_got) <= 0.02 * max(mean_ref, 1.0) + 0.05
    v.close()


@pytest.mark.parametrize('S', [5, 6, 7, 9])
def test_wall_ring_pass_on_the_smallest_maps(S):
    """FireWall on maps whose ring is larger than their interior (4S-4 > (S-2)^2 for S <= 6): the ring pass of the dedicated
    kernel and of the general kernel equal the oracle."""
    from gym_novel_gridworlds_amd import apply_novelty, make_spec
    for fast in ('1', '0'):
        os_env = __import__('os').environ
        old = os_env.get('NGW_FAST_RESET')
        os_env['NGW_FAST_RESET'] = fast
        try:
            spec = make_spec(T.POGO, S)
            spec.items_quantity = {} if S < 6 else ({'crafting_table': 1} if S < 7 else {'crafting_table': 1, 'tree_log': 1})
            apply_novelty(spec, 'firewall', 'medium', '', '')
            n = 500
            v = VecNovelGridworld(spec=spec, num_envs=n, seed=8)
            o = Oracle(spec.compile(), n, seed=8)
            for ep in range(2):
                v.reset(); assert o.reset() == 0
                assert_state_equal(v, o, 'firewall S=%d fast=%s reset %d' % (S, fast, ep))
            v.close()
        finally:
            if old is None:
                del os_env['NGW_FAST_RESET']
            else:
                os_env['NGW_FAST_RESET'] = old
