"""gym.spaces.Box / MultiDiscrete when gym is installed; otherwise minimal local equivalents with the attributes the
reference's callers touch (shape, low, high, nvec, sample, contains)."""
import numpy as np

try:  # pragma: no cover
    import gym
    from gym.spaces import Box, MultiDiscrete
    Env = gym.Env
    HAVE_GYM = True
except Exception:
    HAVE_GYM = False

    class Env:
        metadata = {}
        reward_range = (-float("inf"), float("inf"))
        action_space = None
        observation_space = None

        def close(self):
            pass

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                self.low = np.asarray(low, dtype=dtype)
                self.high = np.asarray(high, dtype=dtype)
            else:
                self.low = np.full(shape, low, dtype=dtype)
                self.high = np.full(shape, high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)
            self.np_random = np.random.RandomState()

        def seed(self, seed=None):
            self.np_random = np.random.RandomState(seed)
            return [seed]

        def sample(self):
            return self.np_random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class MultiDiscrete:
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self.dtype = np.dtype(np.int64)
            self.np_random = np.random.RandomState()

        def seed(self, seed=None):
            self.np_random = np.random.RandomState(seed)
            return [seed]

        def sample(self):
            return (self.np_random.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))
