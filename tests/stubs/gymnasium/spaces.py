import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._np_random = np.random.default_rng(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        super().__init__(shape, dtype, seed)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return self._np_random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))


class Dict(Space):
    def __init__(self, spaces=None, seed=None):
        super().__init__(None, None, seed)
        self.spaces = dict(spaces or {})
