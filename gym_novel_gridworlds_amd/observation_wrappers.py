"""Observation wrappers with the reference's call shape (gym_novel_gridworlds/observation_wrappers.py).

`LidarInFront(env, num_beams=8)` (reference :10-80) works on the single-env adapter (returns the reference's 1-D
np.array of ints from reset()/step()) and on `VecNovelGridworld` (returns a FRESH int32 [N, L] batch per call, like the
reference's fresh np.array - computed in the step launch's own epilogue; the fast path is opt-in: `dtype=np.int16` or
`'packed'` rows and `copy=False`, which hands out the SAME page-locked buffer on every call, overwritten by the next step).  As in the reference, the set of lidar items and the beam range are fixed when the wrapper is
constructed, while the appended inventory follows the env's current items - so a novelty injected AFTER wrapping adds
an inventory entry but no lidar channel (tests/random_action.py:24-42 order).

`AgentMap(env)` (reference :83-129): the 11 x 11 window of the map around the agent (agent_view_size = 5, 0 outside the
map) + agent_facing_id + inventory_items_quantity; the window is gathered by `ngw_agent_view_kernel`."""
import numpy as np

from . import spaces
from .lidar import LidarConfig
from .novelty_wrappers import NoveltyWrapper
from .vec_env import VecNovelGridworld


class LidarInFront(NoveltyWrapper):
    def __init__(self, env, num_beams=8, fused=True, dtype=np.int32, copy=True):
        super().__init__(env)
        self.num_beams = num_beams
        # batched envs: copy=True (default) returns a fresh array per call, as the reference does (a replay buffer may keep it);
        # copy=False hands out the page-locked host rows themselves: ONE buffer, overwritten by the next step()
        self._copy = bool(copy)
        self._dtype = dtype if isinstance(dtype, str) else np.dtype(dtype)   # batched envs: int32 (default), int16 or 'packed' rows (vec_env.lidar_configure)
        self._fused = fused                                     # batched envs: compute the observation inside the step launch
        self._vec = env if isinstance(env, VecNovelGridworld) else None
        spec = env.spec if self._vec is not None else self._base()._sync_spec()
        self._lidar = LidarConfig(spec, num_beams)
        self.lidar_items = set(self._lidar.lidar_items_id)
        self.lidar_items_id = dict(self._lidar.lidar_items_id)
        self.max_beam_range = self._lidar.max_beam_range
        n_inv = len(spec.items) - len(spec.unbreakable_items)
        low = np.array([0] * (len(self.lidar_items) * self.num_beams) + [0] * n_inv)
        high = np.array([self.max_beam_range] * (len(self.lidar_items) * self.num_beams) + [20] * n_inv)
        self.observation_space = spaces.Box(low, high, dtype=int)            # observation_wrappers.py:26-30
        self._configured_for = None

    def _base(self):
        env = self.env
        while isinstance(env, NoveltyWrapper):
            env = env.env
        return env

    def _vec_env(self):
        if self._vec is not None:
            return self._vec
        return self._base()._backend()

    def _ensure(self, vec):
        key = (id(vec), tuple(vec.spec.items_id.items()))
        if self._configured_for != key:
            vec.lidar_configure(self._lidar, fused=self._fused and vec is self._vec, dtype=self._dtype if vec is self._vec else np.int32)
            self._configured_for = key

    def observation(self, obs=None):
        """lidar signal + inventory of the current state (:67-78)."""
        if self._vec is not None:
            self._ensure(self._vec)
            return self._vec.lidar_observation(copy=self._copy)
        base = self._base()
        vec = base._backend()
        base._push(vec)                     # host attributes are the truth between calls (envs.py)
        self._ensure(vec)
        return np.array([int(x) for x in vec.lidar_observation()[0]])

    def reset(self, **kwargs):
        if self._vec is not None:
            self._ensure(self._vec)                              # fused mode must be on before the launch it rides on
        self.env.reset(**kwargs)
        return self.observation()

    def step(self, action):
        if self._vec is not None:
            self._ensure(self._vec)
            _, reward, done, info = self._vec.step(action, with_obs=False)      # the map batch stays on the device
            return self.observation(), reward, done, info
        _, reward, done, info = self.env.step(action)
        return self.observation(), reward, done, info


class AgentMap(NoveltyWrapper):
    def __init__(self, env):
        super().__init__(env)
        self.max_items = 20                                                  # observation_wrappers.py:95-96
        self.agent_view_size = 5
        self._vec = env if isinstance(env, VecNovelGridworld) else None
        items = env.spec.items if self._vec is not None else self.env.items
        assert not self.max_items < len(items), "Cannot have more than " + str(self.max_items) + " items"
        assert self.agent_view_size >= 1, "Increase the agent_view_size"
        # the reference declares (5, 5, 1) although get_agentView returns (11, 11) (:101-102 vs :104-121); kept as declared
        self.observation_space = spaces.Box(low=0, high=self.max_items,
                                            shape=(self.agent_view_size, self.agent_view_size, 1))
        self.observation_space = spaces.Dict({'agent_map': self.observation_space})

    def _base(self):
        env = self.env
        while isinstance(env, NoveltyWrapper):
            env = env.env
        return env

    def get_agentView(self):
        """Local view of the agent (:104-121): single env -> int64 [11, 11] like the reference's np.full window; batched
        env -> int8 [N, 11, 11]."""
        if self._vec is not None:
            return self._vec.agent_view(self.agent_view_size, copy=True)
        base = self._base()
        vec = base._backend()
        base._push(vec)                     # host attributes are the truth between calls (envs.py)
        return vec.agent_view(self.agent_view_size)[0].astype(np.int64)

    def observation(self, obs=None):
        if self._vec is not None:
            cur = self._vec.get_observation() if obs is None else obs
            return {'agent_map': self.get_agentView(), 'agent_facing_id': cur['agent_facing_id'],
                    'inventory_items_quantity': cur['inventory_items_quantity']}
        return {'agent_map': self.get_agentView(), 'agent_facing_id': self.env.agent_facing_id,           # :123-129
                'inventory_items_quantity': self.env.inventory_items_quantity}

    def reset(self, **kwargs):
        obs = self.env.reset(**kwargs)
        return self.observation(obs if self._vec is not None else None)

    def step(self, action):
        if self._vec is not None:
            _, reward, done, info = self._vec.step(action, with_obs=False)      # only the window, pose and inventory travel
            o = self._vec._obs
            return self.observation({'agent_facing_id': o['agent_facing_id'], 'inventory_items_quantity': o['inventory_items_quantity']}), \
                reward, done, info
        obs, reward, done, info = self.env.step(action)
        return self.observation(None), reward, done, info
