"""The O(1) LidarInFront observation on occupancy bit rows (gym_novel_gridworlds_amd/csrc/ngw_boards.inc): the reference's default 8 beams
(observation_wrappers.py:10-80) on maps up to 32 x 32, fused into the in-place step kernel.  Held to the oracle's lidar of the oracle's state
(the oracle's lidar is pinned to vectors captured from the reference wrapper: tests/test_lidar.py) after EVERY launch, through everything
that changes a map: the step's own cell writes, entity pick-ups, prepared-episode copies (whole waves and single lanes, any depth), inline
placements, explicit and masked resets, fused rollouts, graph replays, state injection."""
import numpy as np
import pytest

import ngw_testlib as T
from gym_novel_gridworlds_amd.lidar import LidarConfig
from gym_novel_gridworlds_amd.spec import make_spec

pytestmark = pytest.mark.gpu


def _switched_off():
    """The suite also runs under the A/B switches that take the bit rows away (tools/suite_under_switches.sh): the observations must not change."""
    import os
    return os.environ.get('NGW_LIDAR_BOARDS') == '0' or 'NGW_NOSTAGE' in os.environ or 'NGW_LIDAR_WORLD' in os.environ


def _pair(spec, n, seed, H, prefetch='auto', depth=0, dtype=np.int16, terminal=False):
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle
    v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=seed, autoreset=True, horizon=H, reset_prefetch=prefetch, reset_prefetch_depth=depth,
                            terminal_capture=terminal)
    lc = LidarConfig(spec, 8)
    v.lidar_configure(lc, fused=True, dtype=dtype)
    o = Oracle(spec.compile(), n, seed=seed, autoreset=True, horizon=H)
    return v, o, lc.compile(spec)


def _checker(v, o, cc, spec):
    from oracle.ngw_oracle import lidar
    S, K = spec.map_size, len(spec.items_id)

    def check(where):
        got = v.lidar_observation()
        got = v.lidar_widen(got) if isinstance(got, tuple) else got
        exp = lidar(cc, S, K, o.st.map, o.st.loc, o.st.facing, o.st.inv)
        bad = np.nonzero((got != exp).any(1))[0]
        assert bad.size == 0, (where, bad[:6], got[bad[0]], exp[bad[0]], o.st.loc[bad[0]], o.st.facing[bad[0]])
    return check


CASES = [  # cfg, map size override, envs, horizon, prefetch, depth
    ('pogo10', None, 4000, 17, 'auto', 0), ('pogo10', None, 3000, 9, 0, 0), ('pogo10', None, 2500, 13, 3, 2), ('pogo10', 12, 1500, 15, 4, 0),
    ('pogo13', None, 1200, 11, 'auto', 0), ('pogo10', 14, 900, 19, 5, 0), ('pogo10', 16, 700, 12, 0, 0),
    ('bow20', None, 900, 14, 'auto', 0), ('bow20', 21, 500, 10, 3, 4), ('bow20', 24, 400, 16, 0, 0),
    ('add29h', None, 300, 9, 2, 0), ('add32', None, 300, 8, 'auto', 0), ('add32', None, 200, 7, 0, 0), ('add32', None, 256, 6, 3, 2),
    ('axe10', None, 3000, 25, 'auto', 0), ('bowaxe16', None, 800, 20, 0, 0), ('jump12', None, 1500, 12, 3, 0),
    ('fire10h', None, 2000, 50, 4, 0), ('fire14m', None, 900, 30, 'auto', 0), ('fencer10m', None, 1000, 10, 0, 0), ('fencer12h', None, 800, 9, 2, 0),
    ('crate12h', None, 700, 12, 3, 0), ('fence12h', None, 600, 11, 'auto', 0), ('repl10m', None, 900, 13, 0, 0), ('replwall12e', None, 700, 10, 2, 0),
    ('stk_fire_axe10', None, 1500, 40, 3, 0), ('pogov0_10', None, 1000, 12, 'auto', 0), ('chop10', None, 1000, 15, 2, 0),
]


@pytest.mark.parametrize('cfg,S,n,H,prefetch,depth', CASES)
def test_bit_row_lidar_follows_every_launch(cfg, S, n, H, prefetch, depth):
    import torch
    spec = T.build_spec(cfg, S)
    A = len(spec.actions_id)
    v, o, cc = _pair(spec, n, 23, H, prefetch, depth)
    assert v.step_reads_map_in_place or _switched_off()              # the bit-row path: no map is staged for the fused lidar step
    check = _checker(v, o, cc, spec)
    v.reset(); o.reset(); check('reset')
    stag = (np.arange(n) * 7 % H).astype(np.int32)                      # episode ends spread over the waves: single-lane copies in the cold path
    v.set_state(0, step_count=stag); o.st.step_count[:] = stag
    rs = np.random.RandomState(5)
    for t in range(3 * H + 7):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a)
        check('step %d' % t)
    st = v.get_state()
    assert (st['map'] == o.st.map).all() and (st['inv'] == o.st.inv).all() and (st['loc'] == o.st.loc).all() and (st['episode'] == o.st.episode).all()
    # device-resident steps from a replayed graph (its refills and their bit-row rebuilds are captured with it)
    acts = torch.randint(0, A, (6, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.graph_build(acts.data_ptr(), n, 6); v.graph_launch(3)
    an = acts.cpu().numpy()
    for rep in range(3):
        for t in range(6):
            o.step(an[t])
    check('graph')
    # a fused rollout leaves the bit rows stale; the next step launch rebuilds them
    v.rollout(H + 3, action_seed=3, t0=5); o.rollout(H + 3, 3, 5); check('rollout')
    for t in range(4):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a); check('step after rollout %d' % t)
    mask = (np.arange(n) % 3 == 1).astype(np.uint8)
    v.reset(mask); o.reset(mask); check('masked reset')
    for t in range(H + 2):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a)
    check('steps after the masked reset')
    assert v.error_flags() == 0
    v.close()


@pytest.mark.parametrize('S', [10, 14, 20, 31, 32])
def test_bit_row_lidar_on_injected_maps(S):
    """State injection (ngw_set_state rewrites maps behind the kernels' back) and the corner cases of the ray geometry: an empty interior -
    every ray runs to the ring, a corner-to-corner diagonal is S - 2 cells long and its first range may lie BEYOND max_beam_range (S = 14:
    distance 12 needs range 17, the reference's loop ends at 16 and reports nothing) - and single blocks one cell away in all eight directions."""
    spec = make_spec(T.POGO, S)
    n, K = 512, len(spec.items_id)
    v, o, cc = _pair(spec, n, 4, 0, 0)
    check = _checker(v, o, cc, spec)
    v.reset(); o.reset()
    wall = spec.items_id['wall']
    m = np.zeros((n, S, S), np.int8)
    m[:, 0, :] = wall; m[:, -1, :] = wall; m[:, :, 0] = wall; m[:, :, -1] = wall
    rs = np.random.RandomState(S)
    loc = np.stack([rs.randint(1, S - 1, n), rs.randint(1, S - 1, n)], 1).astype(np.int32)
    loc[:4] = [[1, 1], [S - 2, S - 2], [1, S - 2], [S - 2, 1]]            # the four corners: full-length diagonals
    facing = rs.randint(0, 4, n).astype(np.int32)
    for e in range(8, n):                                               # blocks scattered around (never on the agent)
        for _ in range(rs.randint(0, 12)):
            r, c = rs.randint(1, S - 1, 2)
            if (r, c) != tuple(loc[e]):
                m[e, r, c] = rs.randint(1, K)
    for e, (dy, dx) in enumerate([(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]):
        loc[8 + e] = [S // 2, S // 2]
        m[8 + e, 1:-1, 1:-1] = 0
        m[8 + e, S // 2 + dy, S // 2 + dx] = spec.items_id['tree_log']
    v.set_state(0, map=m.reshape(n, -1), loc=loc, facing=facing)
    o.st.map[:] = m.reshape(n, -1); o.st.loc[:] = loc; o.st.facing[:] = facing
    a = np.full(n, spec.actions_id['Left'], np.int32)                   # a turn: the observation of the injected maps from a new facing
    for t in range(4):
        v.step(a); o.step(a); check('injected, turn %d' % t)
    # other items on the outer ring (every ray still ends there: the ring's bits are forced on, the item of the hit cell is read from the map)
    m2 = o.st.map.reshape(n, S, S).copy()
    m2[::3, 0, 1:-1] = spec.items_id['tree_log']; m2[1::3, 2:-2, S - 1] = spec.items_id['crafting_table']
    v.set_state(0, map=m2.reshape(n, -1)); o.st.map[:] = m2.reshape(n, -1)
    for t in range(4):
        v.step(a); o.step(a); check('other items on the ring, turn %d' % t)
    fw = np.full(n, spec.actions_id['Forward'], np.int32)               # (no Break: a block broken out of the ring would let the oracle's ray leave the map)
    for t in range(6):
        b = fw if t % 2 == 0 else a
        v.step(b); o.step(b); check('other items on the ring, step %d' % t)
    v.close()


@pytest.mark.parametrize('dtype', [np.int32, np.int16, 'packed'])
def test_bit_row_lidar_row_formats_and_one_env_handles(dtype):
    """Every row format, and the handles of at most one wavefront (the gym.Env adapter's: host mirror, sequence word)."""
    for n in (1, 37, 64, 1000):
        spec = T.build_spec('pogo10')
        A = len(spec.actions_id)
        v, o, cc = _pair(spec, n, 9, 11, 'auto', 0, dtype)
        check = _checker(v, o, cc, spec)
        v.reset(); o.reset(); check('reset')
        rs = np.random.RandomState(1)
        for t in range(40):
            a = rs.randint(0, A, size=n).astype(np.int32)
            v.step(a); o.step(a); check('n %d step %d' % (n, t))
        v.close()


def test_bit_row_lidar_with_terminal_capture_and_the_switch():
    """Terminal-observation capture on (the cold path first copies the rows an episode ended in), and NGW_LIDAR_BOARDS=0 restores the march."""
    import os
    spec = T.build_spec('axe10')
    A, n = len(spec.actions_id), 2048
    v, o, cc = _pair(spec, n, 2, 9, 3, 0, np.int16, terminal=True)
    check = _checker(v, o, cc, spec)
    v.reset(); o.reset(); check('reset')
    rs = np.random.RandomState(3)
    for t in range(30):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a); check('step %d' % t)
    v.close()
    os.environ['NGW_LIDAR_BOARDS'] = '0'
    try:
        v, o, cc = _pair(spec, 256, 2, 9, 3)
        assert not v.step_reads_map_in_place
        v.close()
    finally:
        del os.environ['NGW_LIDAR_BOARDS']


@pytest.mark.parametrize('n,slices,dtype', [(20000, '', np.int16), (9000, '3', np.int16), (8192, '', np.int32), (16384, '', 'packed'), (4096, '', np.int16)])
def test_bit_row_lidar_behind_the_pipelined_host_step(n, slices, dtype, monkeypatch):
    """LidarInFront(VecNovelGridworld).step() on a big batch.  Default: the step kernel's write-through form - with a batch of whole wavefronts it
    also stores the observation rows straight into the caller's page-locked buffer (system-scope stores: before they were, a few 64-byte segments
    of 2 MB of rows were still in the GPU's L2 when the launch's sequence number reached the host - the n = 8192 case here caught it), otherwise
    the rows are copied behind the launch.  `slices`: the pipelined form (shifted base pointers for the kernel's hot path, a block offset for its
    cold path), each slice's launch builds its part of the observation rows."""
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle, lidar
    if slices:
        monkeypatch.setenv('NGW_API_SLICES', slices)
    spec = T.build_spec('axe10')
    A, S, K = len(spec.actions_id), spec.map_size, len(spec.items_id)
    v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=6, autoreset=True, horizon=14)
    w = G.LidarInFront(v, num_beams=8, dtype=dtype, copy=False)
    o = Oracle(spec.compile(), n, seed=6, autoreset=True, horizon=14)
    cc = w._lidar.compile(spec)
    wide = (lambda x: v.lidar_widen(x)) if dtype == 'packed' else (lambda x: x)   # (packed rows come back as the pair (beams uint8, inventory int16))
    first = wide(w.reset()); o.reset()
    assert (first == lidar(cc, S, K, o.st.map, o.st.loc, o.st.facing, o.st.inv)).all()
    stag = (np.arange(n) * 5 % 14).astype(np.int32)
    v.set_state(0, step_count=stag); o.st.step_count[:] = stag
    rs = np.random.RandomState(8)
    for t in range(45):
        a = rs.randint(0, A, size=n).astype(np.int32)
        obs, reward, done, info = w.step(a); o.step(a)
        obs = wide(obs)
        assert (reward == o.reward).all() and (done == o.done.astype(bool)).all(), t
        exp = lidar(cc, S, K, o.st.map, o.st.loc, o.st.facing, o.st.inv)
        bad = np.nonzero((obs != exp).any(1))[0]
        assert bad.size == 0, (t, bad[:5])
    assert (v.step_reads_map_in_place or _switched_off()) and v.error_flags() == 0
    st = v.get_state()
    assert (st['map'] == o.st.map).all() and (st['inv'] == o.st.inv).all() and (st['episode'] == o.st.episode).all()
    v.close()
