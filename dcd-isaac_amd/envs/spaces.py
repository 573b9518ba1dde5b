"""Minimal observation/action space descriptors (the reference uses gym.spaces, not installed here).

Only what the reference's callers read: `.shape`, `.n`, `.high`, `.low`, `.dtype`, `.sample()`,
dict-style access on Dict, and `__class__.__name__` in {'Box', 'Discrete', 'Dict'}
(util/__init__.py:135-139, util/make_agent.py:22-47, algos/storage.py:82-93).
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype="float32"):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def sample(self):
        return (np.random.uniform(self.low, self.high, self.shape)).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype("int64")

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __contains__(self, k):
        return k in self.spaces

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def __repr__(self):
        return f"Dict({self.spaces})"
