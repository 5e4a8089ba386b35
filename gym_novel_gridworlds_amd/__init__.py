"""gym_novel_gridworlds_amd - MI355X-native batched step()/reset() hot path of gym-novel-gridworlds.

Scope: NovelGridworld-Pogostick-v1 / Bow-v1 reset+step, the `axe` and `additem` novelties, behind the
reference's gym.Env surface.  Everything computes in hand-written HIP kernels reached through a ctypes C-ABI."""
from .novelty import NOVELTY_NAMES, apply_novelty          # noqa: F401
from .spec import ENV_IDS, STEP_COSTS, EnvSpec, make_spec  # noqa: F401
from .vec_env import VecNovelGridworld                     # noqa: F401

__version__ = '0.1.0'
