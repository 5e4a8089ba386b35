"""LidarInFront observation: host tables + oracle against vectors captured from the reference
(tests/golden/lidar.npz, generator gen_lidar.py); the HIP kernel against the same vectors in -m gpu."""
import json
import os

import numpy as np
import pytest

import ngw_testlib as T
from gym_novel_gridworlds_amd.lidar import LidarConfig
from gym_novel_gridworlds_amd.novelty import apply_novelty
from gym_novel_gridworlds_amd.spec import make_spec

META = json.load(open(os.path.join(T.GOLDEN, 'lidar.json')))
LIDAR = dict(np.load(os.path.join(T.GOLDEN, 'lidar.npz')))


def lidar_setup(cfg):
    """The reference order of tests/random_action.py:24-42: observation wrapper first, novelty injected on top."""
    env_id, S, nov = T.CFGS[cfg]
    spec = make_spec(env_id, S)
    lc = LidarConfig(spec, META[cfg]['num_beams'] if cfg in META else 8)      # configurations without lidar fixtures: the oracle is the checker
    if nov is not None:
        apply_novelty(spec, *nov)
    return spec, lc


def post_states(cfg, n):
    g = T.golden(cfg)
    m = g['ss_pre_map'][:n].copy()
    sel = g['ss_md_c'] < n
    m[g['ss_md_c'][sel], g['ss_md_i'][sel]] = g['ss_md_v'][sel]
    return m, g['ss_post_loc'][:n], g['ss_post_facing'][:n], g['ss_post_inv'][:n]


@pytest.mark.parametrize('cfg', sorted(META))
def test_lidar_tables_match_reference(cfg):
    spec, lc = lidar_setup(cfg)
    ref = META[cfg]
    assert lc.lidar_items_id == ref['lidar_items_id'] and lc.max_beam_range == ref['max_beam_range']
    assert lc.inventory_order(spec) == ref['inventory_order'] and lc.obs_len(spec) == ref['obs_len']
    assert lc.num_beams * len(lc.lidar_items_id) + len(spec.items) - len(spec.unbreakable_items) == ref['obs_len']


@pytest.mark.parametrize('cfg', sorted(META))
def test_oracle_lidar_matches_reference(cfg):
    from oracle.ngw_oracle import lidar
    spec, lc = lidar_setup(cfg)
    n = META[cfg]['n_cases']
    m, loc, facing, inv = post_states(cfg, n)
    got = lidar(lc.compile(spec), spec.map_size, len(spec.items_id), m, loc, facing, inv)
    assert got.shape == LIDAR[cfg + '_obs'].shape
    bad = np.nonzero((got != LIDAR[cfg + '_obs']).any(1))[0]
    assert bad.size == 0, (cfg, bad[:5], got[bad[0]], LIDAR[cfg + '_obs'][bad[0]])
    assert (got[:, :lc.num_beams * len(lc.lidar_items_id)] > 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', sorted(META))
def test_hip_lidar_matches_reference(cfg):
    from gym_novel_gridworlds_amd import VecNovelGridworld
    spec, lc = lidar_setup(cfg)
    n = META[cfg]['n_cases']
    m, loc, facing, inv = post_states(cfg, n)
    v = VecNovelGridworld(spec=spec, num_envs=n)
    v.set_state(0, map=m, loc=loc, facing=facing, inv=inv, selected=np.zeros(n, np.int32), step_count=np.zeros(n, np.int32))
    v.lidar_configure(lc)
    got = v.lidar_observation()
    assert (got == LIDAR[cfg + '_obs']).all()
    dev = v.lidar_observation(device=True)
    assert (dev.cpu().numpy() == got).all()


def _wrapped_env(cfg, backend):
    import gym_novel_gridworlds_amd as G
    env_id, S, nov = T.CFGS[cfg]
    env = G.make(env_id)
    if backend == 'oracle':
        env._make_backend = lambda spec, seed_: T.OracleVec(spec, 1, seed=seed_)
    env.seed(5)
    env.map_size = S
    env = G.LidarInFront(env, num_beams=META[cfg]['num_beams'])     # observation wrapper first ...
    if nov is not None:
        env = G.inject_novelty(env, *nov)                           # ... novelty on top (tests/random_action.py:24-42)
    return env


def _replay_wrapper(cfg, backend, n):
    env = _wrapped_env(cfg, backend)
    base = env.unwrapped if hasattr(env, 'unwrapped') else env
    while hasattr(base, 'env') and not hasattr(base, '_spec'):
        base = base.env
    spec = base._spec
    g = T.golden(cfg)
    first = env.reset()
    assert isinstance(first, np.ndarray) and first.shape == (META[cfg]['obs_len'],)
    for c in range(n):
        T.adapter_inject(base, spec, g['ss_pre_map'][c], g['ss_pre_loc'][c], g['ss_pre_facing'][c], g['ss_pre_sel'][c], g['ss_pre_inv'][c])
        obs, reward, done, info = env.step(int(g['ss_action'][c]))
        assert (obs == LIDAR[cfg + '_obs'][c]).all() and obs.dtype.kind == 'i', (cfg, c)
        assert reward == g['ss_reward'][c] and info['message'] == T.messages()[g['ss_msg'][c]]
    return n


@pytest.mark.parametrize('cfg', ['pogo10', 'axe10', 'bowaxe16'])
def test_lidar_wrapper_on_single_env_adapter(cfg):
    """LidarInFront(env) + inject_novelty on the reference-shaped single env (oracle-backed stand-in backend)."""
    assert _replay_wrapper(cfg, 'oracle', 300) == 300
    env = _wrapped_env(cfg, 'oracle')
    assert list(env.observation_space.shape) == META[cfg]['space_shape']


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', ['pogo10', 'axe10'])
def test_lidar_wrapper_on_hip_backend(cfg):
    assert _replay_wrapper(cfg, 'hip', 120) == 120


@pytest.mark.gpu
def test_lidar_wrapper_on_vec_env_follows_steps():
    """Batched wrapper: observation after every step equals the oracle's lidar of the oracle's state."""
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle, lidar
    spec = T.build_spec('pogo10')
    n = 3000
    v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=8, autoreset=True, horizon=20)
    w = G.LidarInFront(v, num_beams=8)
    o = Oracle(spec.compile(), n, seed=8, autoreset=True, horizon=20)
    first = w.reset()
    o.reset()
    cc = w._lidar.compile(spec)
    assert (first == lidar(cc, 10, 9, o.st.map, o.st.loc, o.st.facing, o.st.inv)).all()
    rs = np.random.RandomState(1)
    for t in range(45):
        a = rs.randint(0, 17, size=n).astype(np.int32)
        obs, reward, done, info = w.step(a)
        o.step(a)
        assert (obs == lidar(cc, 10, 9, o.st.map, o.st.loc, o.st.facing, o.st.inv)).all(), t
        assert (reward == o.reward).all()


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,n,steps,prefetch', [('pogo10', 4000, 60, 0), ('bow20', 700, 40, 0), ('axe10', 2048, 60, 0), ('add32', 128, 30, 0),
                                                  ('pogo10', 3000, 60, 5), ('bow20', 500, 40, 3), ('fire10h', 1500, 50, 4),
                                                  ('fencer10m', 800, 40, 0), ('crate12h', 600, 40, 2),
                                                  # staged maps of more than 64 sixteen-byte chunks: prepared rows reach the LDS copy in several rounds
                                                  ('add36e', 150, 40, 3), ('add36e', 100, 30, 7),
                                                  # 32 x 32: the observation tile shares the candidate masks' LDS (layout_lds), inline and prepared resets
                                                  ('add32', 128, 30, 3)])
def test_fused_lidar_epilogue_matches_oracle(cfg, n, steps, prefetch):
    """ngw_lidar_fuse: reset / step / rollout launches refresh the lidar observation themselves; it equals the oracle's
    lidar of the oracle's state after every launch, and the plain state stays bit-exact too."""
    import torch
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle, lidar
    spec, lc = lidar_setup(cfg)
    A, S, K = len(spec.actions_id), spec.map_size, len(spec.items_id)
    v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=13, autoreset=True, horizon=17, reset_prefetch=prefetch)
    v.lidar_configure(lc, fused=True)
    o = Oracle(spec.compile(), n, seed=13, autoreset=True, horizon=17)
    cc = lc.compile(spec)

    def check(where):
        got = v.lidar_observation()
        exp = lidar(cc, S, K, o.st.map, o.st.loc, o.st.facing, o.st.inv)
        bad = np.nonzero((got != exp).any(1))[0]
        assert bad.size == 0, (where, bad[:4])
        st = v.get_state()
        assert (st['map'] == o.st.map).all() and (st['inv'] == o.st.inv).all() and (st['loc'] == o.st.loc).all()

    v.reset(); o.reset(); check('reset')
    rs = np.random.RandomState(2)
    for t in range(steps):
        a = rs.randint(0, A, size=n).astype(np.int32)
        v.step(a); o.step(a)
        check('step %d' % t)
    acts = torch.randint(0, A, (4, n), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    v.graph_build(acts.data_ptr(), n, 4); v.graph_launch(2)
    an = acts.cpu().numpy()
    for rep in range(2):
        for t in range(4):
            o.step(an[t])
    check('graph')
    v.rollout(23, action_seed=3, t0=5); o.rollout(23, 3, 5); check('rollout')
    mask = (np.arange(n) % 2).astype(np.uint8)
    v.reset(mask); o.reset(mask); check('masked reset')
    v.lidar_configure(lc, fused=False)                     # back to the separate launch
    v.step(an[0]); o.step(an[0]); check('unfused')


def _widen(v, got):
    """lidar_observation() of any row format as one integer [N, L] array."""
    return v.lidar_widen(got) if isinstance(got, tuple) else got


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,beams', [('pogo10', 8), ('bowaxe16', 12), ('pogo13', 5), ('pogo13', 16), ('add32', 4), ('bow20', 6)])
@pytest.mark.parametrize('dtype', [np.int32, np.int16, 'packed'])
def test_lidar_row_formats_and_marches(cfg, beams, dtype, monkeypatch):
    """Every row format (int32, int16, packed uint8 beams + int16 inventory) x both marches - the world-frame one (beam counts
    that are a multiple of 4: wave-uniform ray offsets) and the per-lane table (any other count, or NGW_LIDAR_WORLD=0) - stand-alone
    launch and fused epilogue (step, reset, rollout), against the oracle's lidar of the oracle's state."""
    import gym_novel_gridworlds_amd as G
    from gym_novel_gridworlds_amd.lidar import LidarConfig
    from oracle.ngw_oracle import Oracle, lidar
    spec = T.build_spec(cfg)
    A, S, K = len(spec.actions_id), spec.map_size, len(spec.items_id)
    lc = LidarConfig(spec, beams)
    cc = lc.compile(spec)
    n = 700
    # NGW_LIDAR_WORLD unset: the library's choice (the constant-offset march for the reference's default 8 beams on a 10 x 10 map, else
    # the world-frame one where it applies); 1: the table-driven world-frame march; 0: the per-lane table
    for world in ((None, '1', '0') if beams % 4 == 0 else (None,)):
        if world is None:
            monkeypatch.delenv('NGW_LIDAR_WORLD', raising=False)
        else:
            monkeypatch.setenv('NGW_LIDAR_WORLD', world)
        for fused in (False, True):
            v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=21, autoreset=True, horizon=13)
            v.lidar_configure(lc, fused=fused, dtype=dtype)
            o = Oracle(spec.compile(), n, seed=21, autoreset=True, horizon=13)

            def check(where):
                got = v.lidar_observation()
                if dtype == 'packed':
                    assert got[0].dtype == np.uint8 and got[1].dtype == np.int16 and got[0].shape == (n, beams * len(lc.lidar_items_id))
                else:
                    assert got.dtype == np.dtype(dtype)
                exp = lidar(cc, S, K, o.st.map, o.st.loc, o.st.facing, o.st.inv)
                bad = np.nonzero((_widen(v, got) != exp).any(1))[0]
                assert bad.size == 0, (where, world, fused, bad[:4])
                dev = v.lidar_observation(device=True)
                dev = tuple(x.cpu().numpy() for x in dev) if isinstance(dev, tuple) else dev.cpu().numpy()
                assert (_widen(v, dev) == exp).all(), (where, 'device view')

            v.reset(); o.reset(); check('reset')
            rs = np.random.RandomState(4)
            for t in range(30):
                a = rs.randint(0, A, size=n).astype(np.int32)
                v.step(a); o.step(a)
                if t % 3 == 0 or t > 24:
                    check('step %d' % t)
            v.rollout(17, action_seed=5, t0=2); o.rollout(17, 5, 2); check('rollout')
            mask = (np.arange(n) % 3 == 0).astype(np.uint8)
            v.reset(mask); o.reset(mask); check('masked reset')
            v.close()


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [False, True])
def test_int16_lidar_output(fused):
    """ngw_lidar_set_output(16): the same observation as int16 rows (half the bytes), separate launch and fused epilogue,
    host copy and zero-copy device view; values above 32767 saturate."""
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle, lidar
    spec, lc = lidar_setup('pogo10')
    n = 1000
    v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=3, autoreset=True, horizon=19)
    v.lidar_configure(lc, fused=fused, dtype=np.int16)
    o = Oracle(spec.compile(), n, seed=3, autoreset=True, horizon=19)
    cc = lc.compile(spec)
    v.reset(); o.reset()
    rs = np.random.RandomState(0)
    for t in range(40):
        a = rs.randint(0, 17, size=n).astype(np.int32)
        v.step(a); o.step(a)
        got = v.lidar_observation()
        assert got.dtype == np.int16 and got.shape == (n, lc.obs_len(spec))
        assert (got == lidar(cc, 10, 9, o.st.map, o.st.loc, o.st.facing, o.st.inv)).all(), t
    assert (v.lidar_observation(device=True).cpu().numpy() == got).all()
    st = v.get_state()
    st['inv'][:, spec.items_id['plank']] = 100000                       # beyond int16: saturates instead of wrapping
    v.set_state(0, inv=st['inv'])
    v.step(np.ones(n, np.int32))                                        # a turn: refreshes the fused observation, leaves the inventory alone
    big = v.lidar_observation()
    col = lc.num_beams * len(lc.lidar_items_id) + lc.inventory_order(spec).index('plank')
    assert (big[:, col] == 32767).all()
    w = G.LidarInFront(G.VecNovelGridworld(spec=spec, num_envs=64, seed=1), num_beams=8)
    o1 = w.reset()
    assert o1.dtype == np.int32                                         # the batched wrapper's default: the reference's integers ...
    o2, _, _, _ = w.step(np.zeros(64, np.int32))
    assert not np.shares_memory(o1, o2)                                 # ... in a fresh array per call (a replay buffer may keep them)
    w = G.LidarInFront(G.VecNovelGridworld(spec=spec, num_envs=64, seed=1), num_beams=8, dtype=np.int16, copy=False)
    o1 = w.reset()
    assert o1.dtype == np.int16                                         # the fast path is opt-in: narrow rows, the page-locked buffer itself
    o2, _, _, _ = w.step(np.zeros(64, np.int32))
    assert np.shares_memory(o1, o2)


@pytest.mark.gpu
def test_limit_actions_keeps_a_packed_lidar_setup_and_the_terminal_capture():
    """limit_actions_vec derives a new batched env: the observation setup (incl. the packed row format) and the terminal-observation
    capture travel with it."""
    import gym_novel_gridworlds_amd as G
    from gym_novel_gridworlds_amd.wrappers import limit_actions_vec
    spec = T.build_spec('pogo10')
    v = G.VecNovelGridworld(spec=spec, num_envs=256, seed=1, autoreset=True, horizon=20, terminal_capture=True)
    v.lidar_configure(num_beams=8, fused=True, dtype='packed')
    w = limit_actions_vec(v, {'Forward', 'Left', 'Right', 'Break', 'Craft_plank'})
    assert w.lidar_packed and w.lidar_fused and w.terminal_capture and w.lidar_row_bytes == v.lidar_row_bytes
    w.reset()
    w.step(np.zeros(256, np.int32))
    beams, tail = w.lidar_observation()
    assert beams.dtype == np.uint8 and tail.dtype == np.int16
    v.close(); w.close()
