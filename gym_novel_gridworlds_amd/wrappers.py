"""`SaveTrajectories` (gym_novel_gridworlds/wrappers.py:9-54) and `LimitActions` (:57-85) with the reference's call shape.

    env = LimitActions(env, {'Forward', 'Left', 'Right', 'Break', 'Craft_plank'})

limited id i = the i-th name of sorted(limited_actions); `action_space` shrinks to Discrete(len(limited_actions)).
On the single-env adapter it forwards like the reference wrapper (same AssertionError texts).  On a
`VecNovelGridworld` it compiles the limited table INTO the kernel's action LUT (a new batched env whose action ids are
the limited ids), so there is no per-step translation at all."""
import copy
import os
import pickle
import datetime as _datetime


from . import spaces
from .novelty_wrappers import NoveltyWrapper
from .vec_env import VecNovelGridworld


# What a trajectory entry holds: (key in the pickled dict, attribute of the wrapped env it is read from), in the reference's
# key order (wrappers.py:29-45); the twelfth key, "last_done", is read through the wrapper itself.
_TRAJECTORY_FIELDS = (("map_size", "map_size"), ("map", "map"), ("agent_location", "agent_location"),
                      ("agent_facing_str", "agent_facing_str"), ("block_in_front_id", "block_in_front_id"), ("items_id", "items_id"),
                      ("items_quantity", "items_quantity"), ("inventory_items_quantity", "inventory_items_quantity"),
                      ("action_str", "actions_id"), ("last_action", "last_action"))
_TRAJECTORY_FILE = "%Y-%m-%d-%H-%M-%S"                       # + "_<env_id>.bin" (wrappers.py:49)


class SaveTrajectories(NoveltyWrapper):
    """Reference wrappers.py:9-54: every step appends a snapshot of the env's public state, `save()` pickles the list to
    `<save_path>/<timestamp>_<env_id>.bin`.  Host-side bookkeeping over the single-env adapter: the snapshot's values are the
    live objects (as in the reference, entries alias `env.map` until the env rebinds it)."""

    def __init__(self, env, save_path):
        NoveltyWrapper.__init__(self, env)
        os.makedirs(save_path, exist_ok=True)
        self.save_path, self.state_trajectories = save_path, []

    def get_state(self):
        wrapped = self.env
        snapshot = {key: getattr(wrapped, attribute) for key, attribute in _TRAJECTORY_FIELDS}
        snapshot["last_done"] = self.last_done
        return snapshot

    def step(self, action_id):
        result = self.env.step(action_id)
        self.state_trajectories.append(self.get_state())
        return result

    def save(self):
        name = "%s_%s.bin" % (_datetime.datetime.now().strftime(_TRAJECTORY_FILE), self.env.env_id)
        path = os.path.join(self.save_path, name)
        with open(path, 'wb') as f:
            f.write(pickle.dumps(self.state_trajectories))
        print("Trajectories saved at: ", path)
        return path


# The two AssertionError texts of the reference's LimitActions.step (wrappers.py:76-80); the first one really reads "maxaction"
# (two string literals joined without a space).
_BAD_LIMITED_ID = "Action ID {id} is not valid, maxaction ID is {top}"
_NOT_AN_ACTION = "{name} is not a valid action for {env_id}"


class LimitActions(NoveltyWrapper):
    """Reference wrappers.py:57-85: the agent sees `len(limited_actions)` action ids, limited id i = the i-th name of
    sorted(limited_actions).  A step is two table look-ups - limited id -> name in `limited_actions_id` (the FIRST name
    holding that id, in table order: `remapaction` may install a table of its own), name -> the env's id in `actions_id`,
    read through the wrapper stack so that a remapped or extended action table below is honoured."""

    def __init__(self, env, limited_actions):
        NoveltyWrapper.__init__(self, env)
        self.limited_actions = limited_actions
        self.set_limited_actions_id(dict(zip(sorted(limited_actions), range(len(limited_actions)))))
        self.action_space = spaces.Discrete(len(limited_actions))

    def set_limited_actions_id(self, limited_actions_id):
        self.limited_actions_id = limited_actions_id

    def step(self, action_id):
        table = self.limited_actions_id
        name = next((candidate for candidate, limited in table.items() if limited == action_id), None)
        assert name is not None, _BAD_LIMITED_ID.format(id=action_id, top=len(table) - 1)
        env_actions = self.actions_id
        assert name in env_actions, _NOT_AN_ACTION.format(name=name, env_id=self.env_id)
        return self.env.step(env_actions[name])


def limit_actions_vec(venv, limited_actions):
    """Batched form: a new VecNovelGridworld whose action ids ARE the limited ids (LUT edit, no translation pass)."""
    assert isinstance(venv, VecNovelGridworld)
    spec = copy.deepcopy(venv.spec)
    limited = {action: i for i, action in enumerate(sorted(limited_actions))}
    for action in limited:
        assert action in spec.actions_id, action + " is not a valid action for " + spec.env_id
    spec.actions_id.clear()
    spec.actions_id.update(limited)
    spec.manipulation_actions_id = {a: i for a, i in limited.items() if not a.startswith(('Craft_', 'Select_'))}
    spec.craft_actions_id = {a: i for a, i in limited.items() if a.startswith('Craft_')}
    spec.select_actions_id = {a: i for a, i in limited.items() if a.startswith('Select_')}
    spec.action_space_n = len(limited)
    # (the prepared-episode settings travel as the caller CHOSE them: 'auto' stays adaptive on the derived env)
    new = VecNovelGridworld(spec=spec, num_envs=venv.num_envs, device=venv.device, seed=venv.seed, autoreset=venv.autoreset,
                            horizon=venv.horizon, env_index_base=venv.env_index_base, reset_prefetch=venv._prefetch_arg,
                            reset_prefetch_depth=venv._depth_arg, terminal_capture=venv.terminal_capture)
    if venv.lidar is not None:                              # the observation setup travels with the env
        new.lidar_configure(venv.lidar, fused=venv.lidar_fused, dtype='packed' if venv.lidar_packed else venv.lidar_dtype)
    return new
