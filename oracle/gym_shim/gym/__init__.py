"""Stand-in for the third-party `gym` package (0.18-style surface), TEST INFRASTRUCTURE ONLY.

`gym` is the one dependency of the reference that is absent from this image (no wheel,
no network).  This ~90-line stand-in provides just the names the reference imports so
that `oracle/gen_golden.py` can import the *unmodified* reference from /root/reference
and record golden vectors.  Semantics mirror gym 0.18.0 where they matter to the
reference:

* `Wrapper.__init__` stores `self.env` and COPIES action_space / observation_space /
  reward_range / metadata from the wrapped env,
* `Wrapper.__getattr__` forwards READS only (no `__setattr__` forwarding) and refuses
  `_`-prefixed names,
* `make(id, **kw)` instantiates the entry point with no TimeLimit wrapper.

It is not shipped as part of the product package and nothing under
`gym_novel_gridworlds_amd/` imports it.
"""
from . import error, spaces, utils          # noqa: F401
from .core import Env, Wrapper, ObservationWrapper  # noqa: F401
from . import core, envs                    # noqa: F401
from .envs.registration import make, register  # noqa: F401

__version__ = "0.18.0-standin"
