"""C-ABI argument validation (include/ngw.h): malformed specs and states are rejected with NGW_E_INVALID_ARG and a
message, never launched - the kernels index look-up tables and the map with these values."""
import ctypes as C

import numpy as np
import pytest

import ngw_testlib as T
from gym_novel_gridworlds_amd import VecNovelGridworld, _cabi
from gym_novel_gridworlds_amd.spec import make_spec

pytestmark = pytest.mark.gpu


def _create(cspec, n=64):
    h = C.c_void_p()
    rc = _cabi.lib().ngw_create(C.byref(cspec), n, 0, 0, 0, C.byref(h))
    if rc == 0:
        _cabi.lib().ngw_destroy(h)
    return rc, _cabi.last_error()


@pytest.mark.parametrize('edit,needle', [
    (lambda s: setattr(s, 'goal_item', 30), 'item id out of range'),
    (lambda s: setattr(s, 'n_actions', 99), 'n_actions'),
    (lambda s: s.act_kind.__setitem__(3, 77), 'unknown kind'),
    (lambda s: s.act_arg.__setitem__(8, 7), 'recipe'),                     # action 8 = Craft_*: recipe index out of range
    (lambda s: (setattr(s, 'n_passes', 1), s.pass_kind.__setitem__(0, 1), s.pass_item.__setitem__(0, 3), s.pass_pct_lo.__setitem__(0, 40),
                s.pass_pct_hi.__setitem__(0, 30)), 'additem percent'),
    (lambda s: (setattr(s, 'n_passes', 1), s.pass_kind.__setitem__(0, 2), s.pass_item.__setitem__(0, 3), s.pass_pct_lo.__setitem__(0, 0),
                s.pass_pct_hi.__setitem__(0, 90)), 'replace percent'),
    (lambda s: (setattr(s, 'n_passes', 1), s.pass_kind.__setitem__(0, 7), s.pass_item.__setitem__(0, 3)), 'unknown kind'),
    (lambda s: setattr(s, 'n_passes', 9), 'n_passes'),
    (lambda s: setattr(s, 'fence_mode', 2), 'fence_mode'),
    (lambda s: s.crate_add.__setitem__(2, 3), 'crate_add'),
    (lambda s: (setattr(s, 'n_inv_start', 1), s.inv_start_item.__setitem__(0, 0)), 'inv_start_item'),
    (lambda s: setattr(s, 'map_size', 3), 'map_size'),
    (lambda s: setattr(s, 'abi_version', 99), 'abi'),
    (lambda s: setattr(s, 'n_items', 3), 'n_items'),                       # rows move as 16-byte chunks: at least four items
    (lambda s: (setattr(s, 'n_passes', 2), s.pass_kind.__setitem__(0, 2), s.pass_from.__setitem__(0, s.wall_item), s.pass_item.__setitem__(0, 3),
                s.pass_pct_lo.__setitem__(0, 10), s.pass_pct_hi.__setitem__(0, 20), s.pass_kind.__setitem__(1, 3), s.pass_item.__setitem__(1, 4),
                s.pass_pct_lo.__setitem__(1, 10), s.pass_pct_hi.__setitem__(1, 20)), 'outside the map'),
])
def test_malformed_spec_is_rejected(edit, needle):
    cs = make_spec(T.POGO, 10).compile()
    edit(cs)
    rc, msg = _create(cs)
    assert rc == _cabi.E_INVALID_ARG and needle in msg, (rc, msg)


def test_well_formed_spec_and_bad_counts():
    cs = make_spec(T.POGO, 10).compile()
    assert _create(cs)[0] == 0
    rc, msg = _create(cs, n=0)
    assert rc == _cabi.E_INVALID_ARG and 'n_envs' in msg
    h = C.c_void_p()
    assert _cabi.lib().ngw_create(C.byref(cs), 64, 99, 0, 0, C.byref(h)) in (_cabi.E_INVALID_ARG, _cabi.E_NO_DEVICE)


def test_state_and_call_validation():
    v = VecNovelGridworld(num_envs=100, seed=1)
    v.reset()
    st = v.get_state()
    L = _cabi.lib()
    for key, bad, needle in (('loc', np.array([[0, 3]], np.int32), 'walled interior'), ('loc', np.array([[4, 9]], np.int32), 'walled interior'),
                             ('facing', np.array([4], np.int32), 'agent_facing_id'), ('selected', np.array([9], np.int32), 'selected item'),
                             ('map', np.full((1, 100), 9, np.int8), 'map cell value'), ('inv', np.full((1, 9), -1, np.int32), 'negative')):
        with pytest.raises(ValueError) as ei:
            v.set_state(5, **{key: bad})
        assert needle in str(ei.value), ei.value
    with pytest.raises(ValueError):
        v.set_state(99, facing=np.zeros(2, np.int32))                     # range [99, 101) out of bounds
    after = v.get_state()
    assert all((st[k] == after[k]).all() for k in st)                     # nothing was written by the rejected calls
    for call in (lambda: v.rollout(0), lambda: v.graph_launch(1), lambda: v.lidar_observation(), lambda: v.agent_view(0),
                 lambda: v.set_reset_prefetch(-1), lambda: _cabi.check(L.ngw_set_autoreset(v._h, 1, -5))):
        with pytest.raises(ValueError):
            call()
    from gym_novel_gridworlds_amd.lidar import LidarConfig
    lc = LidarConfig(v.spec, 8).compile(v.spec)
    lc.dr[0][0][0] = 120                                                   # a ray offset far beyond max_range would leave the LDS guard band
    assert L.ngw_lidar_configure(v._h, C.byref(lc)) == _cabi.E_INVALID_ARG and 'max_range' in _cabi.last_error()
    assert L.ngw_reset(None, None) == _cabi.E_INVALID_ARG and 'NULL' in _cabi.last_error()
    v.step(np.zeros(100, np.int32))                                       # the handle is still usable
    assert v.error_flags() == 0


def test_limited_actions_env_keeps_shard_and_observation_setup():
    """limit_actions_vec rebuilds the batched env: global env index base, prepared-episode cadence and the lidar setup travel along."""
    from gym_novel_gridworlds_amd import limit_actions_vec
    v = VecNovelGridworld(num_envs=128, seed=3, autoreset=True, horizon=100, env_index_base=4096, reset_prefetch=16)
    v.lidar_configure(num_beams=8, fused=True, dtype=np.int16)
    w = limit_actions_vec(v, {'Forward', 'Left', 'Right', 'Break', 'Craft_plank'})
    assert (w.env_index_base, w.reset_prefetch, w.lidar_fused, w.lidar_dtype) == (4096, 16, True, np.dtype(np.int16))
    ref = VecNovelGridworld(num_envs=128, seed=3, autoreset=True, horizon=100, env_index_base=4096)
    w.reset(); ref.reset()
    assert all((w.get_state()[k] == ref.get_state()[k]).all() for k in ('map', 'loc', 'facing'))   # same episodes as the shard it came from


def test_inject_novelty_on_a_batched_env_keeps_shard_and_observation_setup():
    """inject_novelty(VecNovelGridworld) rebuilds the SAME env in place (the reference's wrappers mutate the env they wrap,
    novelty_wrappers.py:1586-1674): global env index base, prepared-episode cadence / depth and the lidar setup stay, the old
    handle is closed, and the states equal an oracle batch keyed by the same global indices."""
    import numpy as np
    from gym_novel_gridworlds_amd import inject_novelty
    from oracle.ngw_oracle import Oracle
    import ngw_testlib as T
    v = VecNovelGridworld(num_envs=300, seed=4, env_index_base=7000, autoreset=True, horizon=30, reset_prefetch=7, reset_prefetch_depth=2)
    v.lidar_configure(num_beams=4, fused=True, dtype=np.int16)
    w = inject_novelty(v, 'axe', 'medium', 'wooden', '')
    assert w is v and v._h.value                                          # same object, a live handle (the old one was destroyed)
    assert w.env_index_base == 7000 and w.reset_prefetch == 7 and w.reset_prefetch_depth == 2
    assert w.lidar_fused and w.lidar_dtype == np.dtype(np.int16) and 'wooden_axe' in w.items_id
    spec = T.build_spec('axe10')
    o = Oracle(spec.compile(), 300, seed=4, env_index_base=7000, autoreset=True, horizon=30)
    w.reset(); o.reset()
    rs = np.random.RandomState(1)
    for t in range(70):
        a = rs.randint(0, len(spec.actions_id), size=300).astype(np.int32)
        w.step(a); o.step(a)
    st = w.get_state()
    assert (st['map'] == o.st.map).all() and (st['loc'] == o.st.loc).all() and (st['inv'] == o.st.inv).all() and (st['episode'] == o.st.episode).all()
    assert w.lidar_observation().shape == (300, w.lidar_len)
    w2 = inject_novelty(w, 'breakincrease', 'hard', '', '')          # a stack: still the same object
    assert w2 is v and w2.env_index_base == 7000
    v.close()
