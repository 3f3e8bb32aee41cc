"""Minimal ``gym.spaces.Box`` stand-in (``gym`` is not a dependency of this package).

The reference builds its spaces with ``gym.spaces.Box`` (isaacgymenvs/tasks/base/vec_task.py:102-105);
rl_games only reads ``.shape``, ``.low``, ``.high`` and ``.dtype`` from them.
"""
import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        assert self.low.shape == self.high.shape
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype.name)
