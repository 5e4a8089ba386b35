"""AgentMap observation and SaveTrajectories (reference observation_wrappers.py:83-129, wrappers.py:9-54): the oracle's
numpy restatement against vectors captured from the reference (tests/golden/agentmap.npz, generator gen_agentmap.py);
the HIP gather kernel against the same vectors in -m gpu."""
import os
import pickle

import numpy as np
import pytest

import ngw_testlib as T

GOLD = dict(np.load(os.path.join(T.GOLDEN, 'agentmap.npz')))
CFGS = sorted(k[:-5] for k in GOLD if k.endswith('_view'))


def post_states(cfg, n):
    g = T.golden(cfg)
    m = g['ss_pre_map'][:n].copy()
    sel = g['ss_md_c'] < n
    m[g['ss_md_c'][sel], g['ss_md_i'][sel]] = g['ss_md_v'][sel]
    return m, g['ss_post_loc'][:n], g['ss_post_facing'][:n], g['ss_post_inv'][:n]


@pytest.mark.parametrize('cfg', CFGS)
def test_oracle_agent_view_matches_reference(cfg):
    from oracle.ngw_oracle import agent_view
    n = len(GOLD[cfg + '_view'])
    m, loc, facing, inv = post_states(cfg, n)
    got = agent_view(m, loc, 5)
    assert got.shape == (n, 11, 11) and (got == GOLD[cfg + '_view']).all()
    assert (facing == GOLD[cfg + '_facing']).all() and (inv == GOLD[cfg + '_inv']).all()
    assert (got == 0).any() and (got != 0).any()                      # padding and content both present


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', CFGS)
def test_hip_agent_view_matches_reference(cfg):
    from gym_novel_gridworlds_amd import VecNovelGridworld
    spec = T.build_spec(cfg)
    n = len(GOLD[cfg + '_view'])
    m, loc, facing, inv = post_states(cfg, n)
    v = VecNovelGridworld(spec=spec, num_envs=n)
    v.set_state(0, map=m, loc=loc, facing=facing, inv=inv, selected=np.zeros(n, np.int32), step_count=np.zeros(n, np.int32))
    got = v.agent_view(5)
    assert got.dtype == np.int8 and (got == GOLD[cfg + '_view']).all()
    assert (v.agent_view(5, device=True).cpu().numpy() == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,n,view', [('pogo10', 65536, 5), ('pogo10', 1001, 1), ('bow20', 3000, 12), ('add32', 333, 40)])
def test_hip_agent_view_sizes_match_oracle(cfg, n, view):
    """Other window sizes / batch sizes (ragged tail dword, window larger than the map) against the oracle, after real steps."""
    from gym_novel_gridworlds_amd import VecNovelGridworld
    from oracle.ngw_oracle import Oracle, agent_view
    spec = T.build_spec(cfg)
    v = VecNovelGridworld(spec=spec, num_envs=n, seed=4, autoreset=True, horizon=30)
    o = Oracle(spec.compile(), n, seed=4, autoreset=True, horizon=30)
    v.reset(); o.reset()
    v.rollout(25, action_seed=9); o.rollout(25, 9)
    got = v.agent_view(view)
    assert (got == agent_view(o.st.map, o.st.loc, view)).all()
    with pytest.raises(ValueError):
        v.agent_view(0)


def _wrapped(cfg, backend):
    import gym_novel_gridworlds_amd as G
    env_id, S, nov = T.CFGS[cfg]
    env = G.make(env_id)
    if backend == 'oracle':
        env._make_backend = lambda spec, seed_: T.OracleVec(spec, 1, seed=seed_)
    env.seed(5)
    env.map_size = S
    return G.AgentMap(env)


def _replay(cfg, backend, n):
    env = _wrapped(cfg, backend)
    base = env.env
    first = env.reset()
    assert set(first) == {'agent_map', 'agent_facing_id', 'inventory_items_quantity'}
    spec = base._spec
    g = T.golden(cfg)
    for c in range(n):
        T.adapter_inject(base, spec, g['ss_pre_map'][c], g['ss_pre_loc'][c], g['ss_pre_facing'][c], g['ss_pre_sel'][c], g['ss_pre_inv'][c])
        obs, reward, done, info = env.step(int(g['ss_action'][c]))
        assert obs['agent_map'].shape == (11, 11) and str(obs['agent_map'].dtype) == str(GOLD[cfg + '_dtype'])
        assert (obs['agent_map'] == GOLD[cfg + '_view'][c]).all(), (cfg, c)
        assert obs['agent_facing_id'] == GOLD[cfg + '_facing'][c]
        assert isinstance(obs['inventory_items_quantity'], dict)
        assert [obs['inventory_items_quantity'][k] for k in sorted(base.items_id, key=base.items_id.get)] == list(GOLD[cfg + '_inv'][c])
        assert reward == g['ss_reward'][c]
    assert tuple(env.observation_space.spaces['agent_map'].shape) == tuple(GOLD[cfg + '_space_shape'])
    return n


@pytest.mark.parametrize('cfg', ['pogo10', 'bow20'])
def test_agentmap_wrapper_on_single_env_adapter(cfg):
    assert _replay(cfg, 'oracle', 250) == 250


@pytest.mark.gpu
def test_agentmap_wrapper_on_hip_backend():
    assert _replay('pogo10', 'hip', 100) == 100


@pytest.mark.gpu
def test_agentmap_wrapper_on_vec_env():
    import gym_novel_gridworlds_amd as G
    from oracle.ngw_oracle import Oracle, agent_view
    spec = T.build_spec('pogo10')
    n = 2000
    v = G.VecNovelGridworld(spec=spec, num_envs=n, seed=8, autoreset=True, horizon=20)
    w = G.AgentMap(v)
    o = Oracle(spec.compile(), n, seed=8, autoreset=True, horizon=20)
    first = w.reset(); o.reset()
    assert (first['agent_map'] == agent_view(o.st.map, o.st.loc)).all()
    rs = np.random.RandomState(1)
    for t in range(30):
        a = rs.randint(0, 17, size=n).astype(np.int32)
        obs, reward, done, info = w.step(a)
        o.step(a)
        assert (obs['agent_map'] == agent_view(o.st.map, o.st.loc)).all(), t
        assert (obs['agent_facing_id'] == o.st.facing).all() and (obs['inventory_items_quantity'] == o.st.inv).all()


def test_save_trajectories_wrapper(tmp_path):
    """SaveTrajectories (wrappers.py:9-54): one snapshot per step with the reference's keys; save() pickles the list."""
    import gym_novel_gridworlds_amd as G
    env = G.make('NovelGridworld-Pogostick-v1')
    env._make_backend = lambda spec, seed_: T.OracleVec(spec, 1, seed=seed_)
    env.seed(3)
    env = G.SaveTrajectories(env, str(tmp_path / 'traj'))
    env.reset()
    for a in (0, 1, 2, 3, 0):
        env.step(a)
    assert len(env.state_trajectories) == 5
    st = env.state_trajectories[-1]
    assert set(st) == {"map_size", "map", "agent_location", "agent_facing_str", "block_in_front_id", "items_id", "items_quantity",
                       "inventory_items_quantity", "action_str", "last_action", "last_done"}
    assert st['last_action'] == 'Forward' and st['map_size'] == 10 and st['map'].shape == (10, 10)
    path = env.save()
    assert os.path.basename(path).endswith('_NovelGridworld-Pogostick-v1.bin')
    back = pickle.load(open(path, 'rb'))
    assert len(back) == 5 and back[0]['action_str'] == env.actions_id
