import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
        self._rng = np.random.RandomState(0)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self._rng = np.random.RandomState(0)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(np.int64)
