"""SURVEY §8(f) row 2 - LUT-only novelties and LimitActions: argument errors, wrapper behaviour and the LimitActions
traces against vectors captured from the reference (CPU: oracle-backed stand-in backend; -m gpu: the HIP backend)."""
import numpy as np
import pytest

import ngw_testlib as T
import gym_novel_gridworlds_amd as G
from gym_novel_gridworlds_amd.spec import make_spec
from gym_novel_gridworlds_amd.novelty import apply_novelty

ROW2 = ['brkinc10', 'brkinclog12', 'extdec10', 'axetbe10', 'axetbm12', 'remape10', 'remapm10', 'remaph10', 'chop10', 'jump12',
        'axehard10', 'axehardi12', 'atbhard10', 'atbhardi11']
ROW3 = ['fence10e', 'fence12h', 'fencer10e', 'fencer10m', 'fencer12h', 'repl10m', 'replwall12e', 'fire10h', 'fire14m', 'crate10m',
        'crate12h', 'crate11e']
LIM = dict(np.load(T.GOLDEN + '/limit.npz'))
LIMITED = {'Forward', 'Left', 'Right', 'Break', 'Craft_plank', 'Craft_stick', 'Select_tree_log'}


def test_argument_errors_match_reference():
    for env_id, args, exc, text in T.spec_json()['novelty_arg_errors2']:
        with pytest.raises(Exception) as ei:
            apply_novelty(make_spec(env_id), *args)
        assert type(ei.value).__name__ == exc and str(ei.value) == text, (args, ei.value)
    for args, text in ((('fence', 'hard', '', ''), "For fence novelty, novelty_arg1 (attribute of fence, e.g. oak, jungle) is needed"),
                       (('fencerestriction', 'easy', '', ''), "For fencerestriction novelty, novelty_arg1 (attribute of fence, e.g. oak, jungle) is needed"),
                       (('replaceitem', 'easy', 'wall', ''), "For replaceitem novelty, novelty_arg1 (Item to replace) and novelty_arg2(Item to replace with) are needed"),
                       (('replaceitem', 'easy', 'granite', 'brick'), "Item to replace (granite) is not in the original map"),
                       (('replaceitem', 'easy', 'wall', 'plank'), "Item to replace with (plank) should be a new item"),
                       (('crate', 'tiny', '', ''), "difficulty must be one of 'easy', 'medium', 'hard'")):
        with pytest.raises(AssertionError) as ei:
            apply_novelty(make_spec(T.POGO), *args)
        assert str(ei.value) == text


STACKS = ['stk_axe_bi10', 'stk_bi_axe10', 'stk_add_axe12', 'stk_atb_bi11', 'stk_fen_fire12', 'stk_add_repl12', 'stk_fire_axe10',
          'stk_fire_axeh10', 'stk_fr_axe10', 'stk_axe_fr10', 'stk_crate_fr12', 'stk_fr_crate12', 'stk_crate_bi10']


@pytest.mark.parametrize('cfg', ROW2 + ROW3 + STACKS)
def test_adapter_replays_row2_traces(cfg):
    """inject_novelty on the reference-shaped single env (incl. remapaction reproducing the reference's permutation)."""
    np.random.seed(T.REMAP_SEED.get(cfg, 0))
    assert T.replay_adapter(cfg, 'oracle', max_steps=300, n_single=250) > 400


@pytest.mark.parametrize('cfg', ['axehard10', 'axehardi12', 'atbhard10', 'atbhardi11'])
def test_craftable_axe_wrappers(cfg):
    """AxeHard / AxetoBreakHard (novelty_wrappers.py:216-436, :627-860): only the BASE env's action_space is re-made, the
    ingredients lie on the map (AxeHard) or start in the inventory after every reset (AxetoBreakHard), and the axe is
    crafted at the crafting_table for 6000.0 / +10."""
    ref = T.spec_json()['cfgs'][cfg]
    env = T.make_adapter_env(cfg, 'oracle')
    base = env.env
    nov = T.CFGS[cfg][2]
    axe = nov[2] + '_axe'
    assert env.action_space.n == ref['action_space_n'] and base.action_space.n == ref['base_action_space_n']
    assert base.actions_id == ref['actions_id'] and base.actions_id['Select_' + axe] == len(base.actions_id) - 1
    assert base.inventory_items_quantity[axe] == 0
    recipe = dict(ref['recipes'][axe]['input'])
    if nov[0] == 'axetobreak':
        assert all(base.inventory_items_quantity[k] == q for k, q in recipe.items())     # right after injection (:656)
        assert 'Craft_' + axe not in base.craft_actions_id                                 # :658 leaves that table alone
    env.reset()
    if nov[0] == 'axetobreak':
        assert all(base.inventory_items_quantity[k] == q for k, q in recipe.items())     # and after every reset (:667-670)
        assert not any((base.map == base.items_id[k]).any() for k in recipe)
    else:
        for k, q in recipe.items():
            extra = {'NovelGridworld-Pogostick-v1': {}, 'NovelGridworld-Bow-v1': {}}[base.env_id].get(k, 0)
            assert (base.map == base.items_id[k]).sum() == q + extra
            base.inventory_items_quantity[k] = q
    # face the crafting table and craft the axe
    r, c = base.agent_location
    base.map[r - 1][c] = base.items_id['crafting_table']
    base.set_agent_facing('NORTH')
    obs, reward, done, info = env.step(base.actions_id['Craft_' + axe])
    assert (reward, done, info['step_cost'], info['message']) == (10, False, 6000.0, 'Crafted ' + axe)
    assert base.inventory_items_quantity[axe] == 1 and all(base.inventory_items_quantity[k] == 0 for k in recipe)
    obs, reward, done, info = env.step(base.actions_id['Craft_' + axe])
    assert info['result'] is False and info['step_cost'] == 0 and type(info['step_cost']) is int
    assert info['message'] == 'Missing items: ' + ', '.join('%d %s' % (q, k) for k, q in recipe.items())
    obs, reward, done, info = env.step(base.actions_id['Select_' + axe])
    assert base.selected_item == axe


def _limited_env(kind, backend):
    env = G.make(T.POGO)
    if backend == 'oracle':
        env._make_backend = lambda spec, seed_: T.OracleVec(spec, 1, seed=seed_)
    env.seed(1)
    env = G.LimitActions(env, set(LIMITED))
    if kind == 'remapped':
        np.random.seed(21)
        env = G.inject_novelty(env, 'remapaction', 'hard')
    return env


def _replay_limit(kind, backend, steps=600):
    ref = T.spec_json()['limit_actions'][kind]
    env = _limited_env(kind, backend)
    assert env.limited_actions_id == ref['limited_actions_id'] and env.action_space.n == ref['action_space_n']
    base = env.unwrapped if hasattr(env, 'unwrapped') else env
    while not hasattr(base, '_spec'):
        base = base.env
    env.reset()
    p = 'lim_%s_' % kind
    T.adapter_inject(base, base._spec, LIM[p + 'map0'], LIM[p + 'loc0'], LIM[p + 'facing0'], 0, np.zeros(9, int))
    names = base._spec.item_names
    for t in range(steps):
        if t % 50 == 49:
            base.inventory_items_quantity['tree_log'] += 2
            base.inventory_items_quantity['plank'] += 2
        obs, reward, done, info = env.step(int(LIM[p + 'action'][t]))
        assert reward == LIM[p + 'reward'][t] and done == bool(LIM[p + 'done'][t]) and info['result'] == bool(LIM[p + 'result'][t])
        assert info['step_cost'] == LIM[p + 'cost'][t] and info['message'] == T.messages()[LIM[p + 'msg'][t]], (kind, t)
        assert obs['agent_location'] == tuple(LIM[p + 'loc'][t]) and obs['agent_facing_id'] == LIM[p + 'facing'][t]
        assert [obs['inventory_items_quantity'][n] for n in names] == list(LIM[p + 'inv'][t])
    for a, exc, text in ref['errors']:
        with pytest.raises(AssertionError) as ei:
            env.step(a)
        assert str(ei.value) == text
    return steps


@pytest.mark.parametrize('kind', ['plain', 'remapped'])
def test_limit_actions_wrapper_matches_reference(kind):
    assert _replay_limit(kind, 'oracle') == 600


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['plain', 'remapped'])
def test_limit_actions_wrapper_on_hip_backend(kind):
    assert _replay_limit(kind, 'hip', steps=200) == 200


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', ['brkinc10', 'axetbm12', 'remaph10', 'axehardi12', 'atbhard10', 'fencer10m', 'fire10h', 'crate10m'])
def test_adapter_row2_on_hip_backend(cfg):
    np.random.seed(T.REMAP_SEED.get(cfg, 0))
    assert T.replay_adapter(cfg, 'hip', max_steps=150, n_single=100) > 200


@pytest.mark.gpu
def test_limit_actions_compiled_into_batched_env():
    """limit_actions_vec: the limited ids become the kernel's action table; a batch stepped with limited ids equals the
    full env stepped with the translated ids."""
    from oracle.ngw_oracle import Oracle
    n = 2000
    full = G.VecNovelGridworld(num_envs=n, seed=4, autoreset=True, horizon=30)
    lim = G.limit_actions_vec(full, LIMITED)
    assert lim.action_space.n == 7 and lim.actions_id == {a: i for i, a in enumerate(sorted(LIMITED))}
    o = Oracle(full.spec.compile(), n, seed=4, autoreset=True, horizon=30)
    lim.reset(); o.reset()
    table = np.array([full.actions_id[a] for a in sorted(LIMITED)], np.int32)
    rs = np.random.RandomState(0)
    for t in range(80):
        a = rs.randint(0, 7, size=n).astype(np.int32)
        _, reward, done, info = lim.step(a)
        o.step(table[a])
        assert (reward == o.reward).all() and (info['message_code'] == o.msg_code).all()
    st = lim.get_state()
    assert (st['map'] == o.st.map).all() and (st['inv'] == o.st.inv).all()
    with pytest.raises(ValueError):
        lim.step(np.full(n, 7, np.int32))


def test_novelty_over_limit_actions_asserts_like_reference():
    """A novelty wrapper stacked ON a LimitActions wrapper checks on every step that the action it overrides survived the
    limiting (novelty_wrappers.py:40, :139, :264-265, :467, :913, :1080, :1283, :1429, :1510)."""
    def limited(actions, nov):
        env = G.make(T.POGO if nov[0] != 'extractincdec' else T.BOW)
        env._make_backend = lambda spec, seed_: T.OracleVec(spec, 1, seed=seed_)
        env.seed(1)
        env = G.inject_novelty(G.LimitActions(env, set(actions)), *nov)
        env.reset()
        return env
    cases = [(('axe', 'medium', 'wooden'), "Cannot use breakincrease novelty_arg2 because you do not have Break in LimitActions"),
             (('axetobreak', 'easy', 'iron'), "Cannot use axetobreak novelty because you do not have Break in LimitActions"),
             (('breakincrease', 'hard'), "Cannot use breakincrease novelty because you do not have Break in LimitActions"),
             (('crate', 'easy'), "Cannot use crate novelty because you do not have Break in LimitActions"),
             (('fencerestriction', 'hard', 'oak'), "Cannot use fencerestriction novelty because you do not have Break in LimitActions"),
             (('addchop', 'hard'), "Cannot use addchop novelty because you do not have Chop in LimitActions"),
             (('axe', 'hard', 'wooden'), "Cannot use AxeHard novelty because you do not have Craft_wooden_axe in LimitActions")]
    for nov, text in cases:
        env = limited(['Forward', 'Left', 'Right'], nov)
        with pytest.raises(AssertionError) as ei:
            env.step(0)
        assert str(ei.value) == text, (nov, ei.value)
    env = limited(['Forward', 'Left', 'Right'], ('extractincdec', 'hard', 'decrease'))
    with pytest.raises(AssertionError) as ei:
        env.step(0)
    assert str(ei.value) == "Cannot use extractincdec novelty because you do not have Extract action in LimitActions"
    env = limited(['Forward', 'Left', 'Right', 'Break'], ('axe', 'medium', 'wooden'))        # requirement met: steps go through
    obs, reward, done, info = env.step(env.limited_actions_id['Break'])
    assert info['step_cost'] == 3600.0


def test_fence_after_a_wall_replacing_novelty_is_refused():
    """firewall / replaceitem(wall -> X) below fence / fencerestriction: the reference's add_fence_around indexes outside the map
    (numpy wraps row -1 around and raises IndexError on row S); the stack is refused when it is built, nothing is launched."""
    from gym_novel_gridworlds_amd import apply_novelty, make_spec
    for first in (('firewall', 'hard', '', ''), ('replaceitem', 'medium', 'wall', 'brick')):
        for second in (('fence', 'hard', 'oak', ''), ('fencerestriction', 'medium', 'oak', '')):
            spec = make_spec('NovelGridworld-Pogostick-v1', 10)
            apply_novelty(spec, *first)
            with pytest.raises(IndexError):
                apply_novelty(spec, *second)
    spec = make_spec('NovelGridworld-Pogostick-v1', 10)                      # the other order is fine (fences first: no border cell is fenced)
    apply_novelty(spec, 'fence', 'hard', 'oak', '')
    apply_novelty(spec, 'firewall', 'hard', '', '')
    spec.compile()
