"""Action/observation space descriptors.

The reference declares `spaces.Discrete(len(positions))` and
`spaces.Box(-inf, inf, shape=[...])` (environments.py:112-123).  Gymnasium's own
classes are used when gymnasium is importable; otherwise these minimal stand-ins
with the same attributes (n / nvec / low / high / shape / dtype, sample, contains).
"""
from __future__ import annotations

import numpy as np

try:  # soft dependency: not installed in the build image
    from gymnasium.spaces import Box, Discrete, MultiDiscrete  # type: ignore  # noqa: F401
    HAVE_GYMNASIUM = True
except Exception:  # pragma: no cover - exercised when gymnasium is absent
    HAVE_GYMNASIUM = False

    class Discrete:
        def __init__(self, n, seed=None):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(self._rng.integers(0, self.n))

        def contains(self, x):
            return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete:
        def __init__(self, nvec, seed=None):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self.dtype = np.int64
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.integers(0, self.nvec)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(((x >= 0) & (x < self.nvec)).all())

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.shape = tuple(int(s) for s in shape)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.standard_normal(self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(((x >= self.low) & (x <= self.high)).all())

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
