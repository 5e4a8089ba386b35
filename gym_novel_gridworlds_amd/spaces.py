"""Action / observation space objects.  Uses `gym.spaces` when classic gym is importable, else minimal stand-ins
with the attributes the reference's users touch (`Discrete.n`, `.sample()`, `Box.shape`, `Dict.spaces`)."""
import numpy as np

try:                                                    # classic gym (0.18-style API), as the reference uses
    from gym import spaces as _gs
    Discrete, Box, Dict = _gs.Discrete, _gs.Box, _gs.Dict
    HAVE_GYM = True
except Exception:                                       # noqa: BLE001 - gym absent (or broken): stand-ins
    HAVE_GYM = False

    class Discrete(object):
        def __init__(self, n):
            self.n = int(n)
            self.shape, self.dtype = (), np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def __repr__(self):
            return "Box" + str(self.shape)

    class Dict(object):
        def __init__(self, spaces=None):
            self.spaces = dict(spaces or {})

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def __repr__(self):
            return "Dict(" + ", ".join(k + ":" + repr(s) for k, s in self.spaces.items()) + ")"
