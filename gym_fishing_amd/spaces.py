"""Minimal Box / Discrete spaces (base_fishing_env.py:49-58, fishing_env.py:23-24).

`gym` / `gymnasium` are optional: when one is importable its space classes are used so
the envs plug into SB3-style tooling; otherwise these stand-alone equivalents carry the
same attributes (low, high, shape, dtype, n, sample, contains).  Never a hard import.
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.low = np.asarray(low, dtype=self.dtype)
        self.high = np.asarray(high, dtype=self.dtype)
        if shape is not None:
            self.low = np.broadcast_to(self.low, shape).copy()
            self.high = np.broadcast_to(self.high, shape).copy()
        self.shape = self.low.shape
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        try:
            xi = int(x)
        except (TypeError, ValueError):
            return False
        return 0 <= xi < self.n and xi == x

    __contains__ = contains

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n


def _external_spaces():
    for mod in ("gymnasium", "gym"):
        try:
            m = __import__(mod + ".spaces", fromlist=["Box", "Discrete"])
            return m.Box, m.Discrete
        except Exception:  # noqa: BLE001 - optional dependency, any failure means "absent"
            continue
    return None


def space_classes():
    """(Box, Discrete) from gymnasium / gym when importable, else the stand-alone ones."""
    return _external_spaces() or (Box, Discrete)


def is_discrete(space):
    return hasattr(space, "n") and not hasattr(space, "low")
