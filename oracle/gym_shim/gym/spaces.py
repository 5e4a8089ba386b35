import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Discrete(Space):
    def __init__(self, n):
        assert n >= 0
        self.n = n
        super().__init__((), np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        super().__init__(shape, dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return "Box" + str(self.shape)


class Dict(Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})
        super().__init__(None, None)

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def __repr__(self):
        return "Dict(" + ", ".join(k + ":" + repr(s) for k, s in self.spaces.items()) + ")"
