"""`inject_novelty` with the reference's call shape (gym_novel_gridworlds/novelty_wrappers.py:1586-1674).

    env = inject_novelty(env, novelty_name, difficulty='hard', novelty_arg1='', novelty_arg2='')

Accepts a single-env adapter (envs.py) - returns a wrapper that forwards reads to the wrapped env exactly like
`gym.core.Wrapper` (copies action_space / observation_space at construction, `__getattr__` for reads only) - or a
`VecNovelGridworld` / `ShardedVecNovelGridworld`, which is rebuilt IN PLACE on the edited spec and returned (the
reference's wrappers mutate the env they wrap; a shard keeps its place in the global env index space)."""
from .novelty import apply_novelty


class NoveltyWrapper(object):
    """Reads fall through to the wrapped env; writes stay on the wrapper (gym 0.18 `Wrapper` semantics)."""

    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space                 # copied, NOT grown by axe/additem
        self.observation_space = env.observation_space
        self.reward_range = getattr(env, 'reward_range', (-float('inf'), float('inf')))
        self.metadata = getattr(env, 'metadata', {})

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, 'unwrapped', self.env)

    # (action name that must survive a LimitActions wrapper BELOW this one, assertion text) - the reference's step()
    # overrides check these on every step when `limited_actions_id` is visible through the wrapped env
    _limit_requirements = ()

    def step(self, action_id):
        if self._limit_requirements and hasattr(self, 'limited_actions_id'):
            limited = self.limited_actions_id
            for name, message in self._limit_requirements:
                ok = any(a.startswith(name[:-1]) for a in limited) if name.endswith('*') else name in limited
                assert ok, message
        return self.env.step(action_id)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def render(self, mode='human', **kwargs):
        return self.env.render(mode, **kwargs)

    def close(self):
        return self.env.close()

    def seed(self, seed=None):
        return self.env.seed(seed)


class AxeEasy(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use breakincrease novelty_arg2 because you do not have Break in LimitActions"),)   # :40


class AxeMedium(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use breakincrease novelty_arg2 because you do not have Break in LimitActions"),)   # :139


class AxeHard(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use breakincrease novelty_arg2 because you do not have Break in LimitActions"),)   # :265 (+ Craft_<axe>, set per instance)


class AxetoBreakHard(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use axetobreak novelty because you do not have Break in LimitActions"),)   # :680

    def reset(self):                                          # novelty_wrappers.py:664 takes no kwargs
        return self.env.reset()


class AxetoBreakEasy(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use axetobreak novelty because you do not have Break in LimitActions"),)   # :467


class AxetoBreakMedium(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use axetobreak novelty because you do not have Break in LimitActions"),)   # :557


class BreakIncrease(NoveltyWrapper):
    _limit_requirements = (('Break', "Cannot use breakincrease novelty because you do not have Break in LimitActions"),)   # :1429


class ExtractIncDec(NoveltyWrapper):
    _limit_requirements = (('Extract*', "Cannot use extractincdec novelty because you do not have Extract action in LimitActions"),)   # :1504-1510


class AddChopAction(NoveltyWrapper):
    _limit_requirements = (('Chop', "Cannot use addchop novelty because you do not have Chop in LimitActions"),)   # :1283


class AddJumpAction(NoveltyWrapper):
    _limit_requirements = (('Jump', "Cannot use addjump novelty because you do not have Jump in LimitActions"),)   # :1355


class AddItem(NoveltyWrapper):
    def reset(self):                                          # novelty_wrappers.py:1013 takes no kwargs
        return self.env.reset()


class Fence(AddItem):                                         # reset() without kwargs: :867
    pass


class FenceRestriction(AddItem):                              # :902
    _limit_requirements = (('Break', "Cannot use fencerestriction novelty because you do not have Break in LimitActions"),)   # :913


class Crate(AddItem):                                         # :1070
    _limit_requirements = (('Break', "Cannot use crate novelty because you do not have Break in LimitActions"),)   # :1080


class ReplaceItem(AddItem):                                   # :1128
    pass


class FireWall(AddItem):                                      # :1161
    pass


def inject_novelty(env, novelty_name, difficulty='hard', novelty_arg1='', novelty_arg2=''):
    if callable(getattr(env, 'rebuild', None)):
        # batched env (or a rank-local shard of one, dist.py): the edited spec is compiled into the kernels' tables IN PLACE -
        # same object, same global env indices, same autoreset / prepared-episode / lidar settings (VecNovelGridworld.rebuild)
        import copy
        spec = copy.deepcopy(env.spec)
        apply_novelty(spec, novelty_name, difficulty, novelty_arg1, novelty_arg2)
        return env.rebuild(spec)
    base = getattr(env, 'unwrapped', env)
    base = getattr(base, 'env', base) if isinstance(base, NoveltyWrapper) else base
    spec = base._sync_spec()
    if novelty_name == 'remapaction' and hasattr(env, 'limited_actions_id'):
        # remap_action_difficulty with LimitActions in the stack: only the limited table is shuffled (:1209-1210)
        assert difficulty in ['easy', 'medium', 'hard'], "difficulty must be one of 'easy', 'medium', 'hard'"
        from .novelty import _remap_action
        env.set_limited_actions_id(_remap_action(env.limited_actions_id, 0))
        return env
    apply_novelty(spec, novelty_name, difficulty, novelty_arg1, novelty_arg2)     # validates like the reference
    for name in ('manipulation_actions_id', 'craft_actions_id', 'select_actions_id'):
        setattr(base, name, getattr(spec, name))                                  # remapaction re-binds these tables
    if novelty_name in ('axe', 'axetobreak') and difficulty == 'hard':
        from . import spaces
        had_iron = 'iron' in base.inventory_items_quantity or novelty_arg1 != 'iron'
        w = AxeHard(env) if novelty_name == 'axe' else AxetoBreakHard(env)     # copies the OLD action_space (gym.Wrapper.__init__)
        craft = 'Craft_' + novelty_arg1 + '_axe'                               # :263-264 / :678-679: checked before Break
        w._limit_requirements = ((craft, "Cannot use " + type(w).__name__ + " novelty because you do not have " + craft +
                                  " in LimitActions"),) + type(w)._limit_requirements
        base.action_space = spaces.Discrete(len(base.actions_id))              # :256 / :662 re-make the base env's space
        base.inventory_items_quantity.update({novelty_arg1 + '_axe': 0})       # :230 / :643
        if novelty_name == 'axetobreak':
            base.inventory_items_quantity.update(spec.start_inventory)         # :656 - ingredients in the inventory right away
        elif not had_iron:
            base.reset()            # AxeHard.__init__ -> add_new_items({'iron': 3}) -> reset() (:250)
        return w
    if novelty_name == 'axe':
        if difficulty == 'medium':
            base.reset()            # AxeMedium.__init__ -> add_new_items -> reset(): the axe appears on the map (:129)
            return AxeMedium(env)
        base.inventory_items_quantity.update({novelty_arg1 + '_axe': 1})          # AxeEasy.__init__ :22
        return AxeEasy(env)
    if novelty_name == 'axetobreak':
        if difficulty == 'medium':
            base.reset()            # AxetoBreakMedium.__init__ -> add_new_items -> reset() (:551)
            return AxetoBreakMedium(env)
        base.inventory_items_quantity.update({novelty_arg1 + '_axe': 1})          # AxetoBreakEasy.__init__ :451
        return AxetoBreakEasy(env)
    if novelty_name == 'breakincrease':
        return BreakIncrease(env)
    if novelty_name == 'extractincdec':
        return ExtractIncDec(env)
    if novelty_name == 'remapaction':
        return env                  # remap_action_difficulty returns the env itself (:1227)
    if novelty_name in ('addchop', 'addjump'):
        w = AddChopAction(env) if novelty_name == 'addchop' else AddJumpAction(env)
        from . import spaces
        w.action_space = spaces.Discrete(len(base.actions_id))     # these two wrappers DO grow their action_space (:1278, :1350)
        return w
    return {'additem': AddItem, 'fence': Fence, 'fencerestriction': FenceRestriction, 'crate': Crate,
            'replaceitem': ReplaceItem, 'firewall': FireWall}[novelty_name](env)
