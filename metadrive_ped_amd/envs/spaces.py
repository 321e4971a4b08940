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


class LazyInfo(dict):
    """The `info` of a batched step: a dict whose derived entries (flag bits -> bool tensors ...) are computed when they
    are first READ.  Building all of them eagerly costs some thirty small kernel launches per step -- several times the
    step itself -- while a training loop reads two or three.  Like the entries that are plain views of the engine's
    buffers, a derived entry reflects the state at the moment it is read: read (or clone) what you need before the next
    step."""
    def __init__(self, eager, lazy):
        super().__init__(eager)
        self._lazy = dict(lazy)

    def __missing__(self, key):
        fn = self._lazy.pop(key)            # KeyError for an unknown key, like a dict
        value = fn()
        self[key] = value
        return value

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def __iter__(self):
        yield from dict.__iter__(self)
        yield from list(self._lazy)

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)

    def keys(self):
        return list(iter(self))

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def get(self, key, default=None):
        return self[key] if key in self else default
