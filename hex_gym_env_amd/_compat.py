"""gymnasium is optional: the reference imports it at module top (HexGame.py:1-2); when it is
absent (as in the build image) minimal stand-ins keep the same attribute surface."""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as gym
    from gymnasium import spaces
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium not installed
    HAVE_GYMNASIUM = False

    class _Env:
        metadata = {}

        def close(self):
            pass

    class _Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(((x >= self.low) & (x <= self.high)).all())

        def sample(self):
            return np.random.randint(self.low, self.high + 1, size=self.shape).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class _Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def contains(self, x):
            return 0 <= int(x) < self.n

        def sample(self):
            return int(np.random.randint(self.n))

        def __repr__(self):
            return f"Discrete({self.n})"

    class _Spaces:
        Box = _Box
        Discrete = _Discrete

    class _Gym:
        Env = _Env
        spaces = _Spaces

    gym = _Gym
    spaces = _Spaces
