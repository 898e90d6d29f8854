"""`Box` space: the real gymnasium.spaces.Box when gymnasium is importable, otherwise a small
stand-in with the attributes the reference's controllers read (`.shape .low .high .dtype
.sample() .seed() .contains()`; e.g. controllers/ppo/ppo_utils.py:213 `isinstance(act_space, Box)`)."""
import numpy as np

try:                                            # pragma: no cover - gymnasium is absent in this image
    from gymnasium.spaces import Box            # noqa: F401
    HAVE_GYMNASIUM = True
except Exception:                               # noqa: BLE001
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low) if np.ndim(low) else np.shape(high)
            self._shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self._shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self._shape).copy()
            self._np_random = None
            if seed is not None:
                self.seed(seed)

        @property
        def shape(self):
            return self._shape

        def seed(self, seed=None):
            self._np_random = np.random.default_rng(seed)
            return [seed]

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed()
            return self._np_random

        def sample(self):
            return self.np_random.uniform(self.low, self.high, size=self._shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return bool(x.shape == self._shape and np.all(x >= self.low) and np.all(x <= self.high))

        def __contains__(self, x):
            return self.contains(x)

        def __repr__(self):
            return f'Box({self.low}, {self.high}, {self._shape}, {self.dtype})'
