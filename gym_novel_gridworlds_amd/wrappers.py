"""`SaveTrajectories` (gym_novel_gridworlds/wrappers.py:9-54) and `LimitActions` (:57-85) with the reference's call shape.

    env = LimitActions(env, {'Forward', 'Left', 'Right', 'Break', 'Craft_plank'})

limited id i = the i-th name of sorted(limited_actions); `action_space` shrinks to Discrete(len(limited_actions)).
On the single-env adapter it forwards like the reference wrapper (same AssertionError texts).  On a
`VecNovelGridworld` it compiles the limited table INTO the kernel's action LUT (a new batched env whose action ids are
the limited ids), so there is no per-step translation at all."""
import copy
import os
import pickle
from datetime import datetime


from . import spaces
from .novelty_wrappers import NoveltyWrapper
from .vec_env import VecNovelGridworld


class SaveTrajectories(NoveltyWrapper):
    """Reference wrappers.py:9-54: after every step append a snapshot of the env's state; `save()` pickles the list to
    `<save_path>/<timestamp>_<env_id>.bin`.  Host-side bookkeeping over the single-env adapter's attributes (the values
    are references to the live objects, exactly as in the reference, so entries alias `env.map` until it is rebound)."""

    def __init__(self, env, save_path):
        super().__init__(env)
        self.save_path = save_path
        os.makedirs(self.save_path, exist_ok=True)
        self.state_trajectories = []

    def step(self, action_id):
        obs, reward, done, info = self.env.step(action_id)
        self.state_trajectories.append(self.get_state())
        return obs, reward, done, info

    def get_state(self):
        env = self.env
        return {"map_size": env.map_size, "map": env.map, "agent_location": env.agent_location,           # :29-45
                "agent_facing_str": env.agent_facing_str, "block_in_front_id": env.block_in_front_id,
                "items_id": env.items_id, "items_quantity": env.items_quantity,
                "inventory_items_quantity": env.inventory_items_quantity,
                "action_str": env.actions_id, "last_action": env.last_action, "last_done": self.last_done}

    def save(self):
        path = os.path.join(self.save_path,
                            datetime.now().strftime("%Y-%m-%d-%H-%M-%S") + "_{env}.bin".format(env=self.env.env_id))
        with open(path, 'wb') as f:
            pickle.dump(self.state_trajectories, f)
        print("Trajectories saved at: ", path)
        return path


class LimitActions(NoveltyWrapper):
    def __init__(self, env, limited_actions):
        super().__init__(env)
        self.limited_actions = limited_actions
        self.limited_actions_id = {action: i for i, action in enumerate(sorted(self.limited_actions))}   # wrappers.py:67
        self.action_space = spaces.Discrete(len(self.limited_actions_id))

    def set_limited_actions_id(self, limited_actions_id):
        self.limited_actions_id = limited_actions_id

    def step(self, action_id):
        assert action_id in self.limited_actions_id.values(), "Action ID " + str(action_id) + " is not valid, max" \
                                                              "action ID is " + str(len(self.limited_actions_id) - 1)
        last_action = list(self.limited_actions_id.keys())[list(self.limited_actions_id.values()).index(action_id)]
        assert last_action in self.actions_id, last_action + " is not a valid action for " + self.env_id
        action_id = self.actions_id[last_action]
        return self.env.step(action_id)


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
                            reset_prefetch_depth=venv._depth_arg)
    if venv.lidar is not None:                              # the observation setup travels with the env
        new.lidar_configure(venv.lidar, fused=venv.lidar_fused, dtype=venv.lidar_dtype)
    return new
