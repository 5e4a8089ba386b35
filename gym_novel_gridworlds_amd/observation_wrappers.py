"""Observation wrappers with the reference's call shape (gym_novel_gridworlds/observation_wrappers.py).

`LidarInFront(env, num_beams=8)` (reference :10-80) works on the single-env adapter (returns the reference's 1-D
np.array of ints from reset()/step()) and on `VecNovelGridworld` (returns int32 [N, L] batches computed by the
`ngw_lidar_kernel`).  As in the reference, the set of lidar items and the beam range are fixed when the wrapper is
constructed, while the appended inventory follows the env's current items - so a novelty injected AFTER wrapping adds
an inventory entry but no lidar channel (tests/random_action.py:24-42 order)."""
import numpy as np

from . import spaces
from .lidar import LidarConfig
from .novelty_wrappers import NoveltyWrapper
from .vec_env import VecNovelGridworld


class LidarInFront(NoveltyWrapper):
    def __init__(self, env, num_beams=8, fused=True):
        super().__init__(env)
        self.num_beams = num_beams
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
            vec.lidar_configure(self._lidar, fused=self._fused and vec is self._vec)
            self._configured_for = key

    def observation(self, obs=None):
        """lidar signal + inventory of the current state (:67-78)."""
        if self._vec is not None:
            self._ensure(self._vec)
            return self._vec.lidar_observation(copy=True)
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
        _, reward, done, info = self.env.step(action)
        return self.observation(), reward, done, info
