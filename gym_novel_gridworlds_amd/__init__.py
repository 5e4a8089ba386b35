"""gym_novel_gridworlds_amd - MI355X-native batched step()/reset() hot path of gym-novel-gridworlds.

Scope: NovelGridworld-Pogostick-v1 / Bow-v1 reset+step, the `axe` and `additem` novelties, behind the
reference's gym.Env surface.  Everything computes in hand-written HIP kernels reached through a ctypes C-ABI.

    import gym, gym_novel_gridworlds_amd                    # registers the ids when classic gym is importable
    env = gym.make('NovelGridworld-Pogostick-v1')           # or gym_novel_gridworlds_amd.make(...)
    venv = gym_novel_gridworlds_amd.VecNovelGridworld('NovelGridworld-Pogostick-v1', num_envs=65536)
"""
from .envs import ENTRY_POINTS, BowV0Env, BowV1Env, PogostickV0Env, PogostickV1Env, make   # noqa: F401
from .novelty import NOVELTY_NAMES, apply_novelty                    # noqa: F401
from .novelty_wrappers import inject_novelty                         # noqa: F401
from .observation_wrappers import AgentMap, LidarInFront                    # noqa: F401
from .spec import ENV_IDS, STEP_COSTS, EnvSpec, make_spec            # noqa: F401
from .vec_env import VecNovelGridworld                               # noqa: F401
from .wrappers import LimitActions, SaveTrajectories, limit_actions_vec             # noqa: F401

__version__ = '0.1.0'


def register_with_gym():
    """register(id, entry_point) for the two in-scope ids (gym_novel_gridworlds/__init__.py:47-60); no kwargs,
    no max_episode_steps.  Returns the ids registered (none when classic gym is not importable)."""
    try:
        from gym.envs.registration import register
    except Exception:                                       # noqa: BLE001 - gym is optional
        return []
    done = []
    for env_id, cls in ENTRY_POINTS.items():
        try:
            register(id=env_id, entry_point='gym_novel_gridworlds_amd.envs:' + cls.__name__)
            done.append(env_id)
        except Exception:                                   # noqa: BLE001 - already registered (e.g. by the reference)
            pass
    return done


REGISTERED_IDS = register_with_gym()
