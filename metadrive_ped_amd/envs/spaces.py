"""Minimal Box / Discrete / MultiDiscrete spaces (gymnasium is not a dependency of the batched engine).  Same fields and checks
as gymnasium.spaces.Box for the uses the reference makes of it (obs/state_obs.py:24-28,172-183,
policy/env_input_policy.py:61-62)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __repr__(self):
        return "Box({}, {}, {}, {})".format(self.low.min(), self.high.max(), self.shape, self.dtype)


class Discrete:
    """gymnasium.spaces.Discrete as used by EnvInputPolicy.get_input_space (policy/env_input_policy.py:64-68)."""
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return int(self._rng.randint(self.n))

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == () and np.issubdtype(x.dtype, np.integer) and 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete({})".format(self.n)


class MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.issubdtype(x.dtype, np.integer) and bool(np.all(x >= 0)) and \
            bool(np.all(x < self.nvec))

    def __repr__(self):
        return "MultiDiscrete({})".format(self.nvec.tolist())
