"""NumPy restatement of the reference's running normalisers — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/safe_control_gym/math_and_models/normalization.py:13-55 (RunningMeanStd), :88-124
(MeanStdNormalizer), :127-159 (RewardStdNormalizer, including the `ret[dones.astype(long)] = 0` indexing at :158).
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean, self.var, self.count = np.zeros(shape, np.float64), np.ones(shape, np.float64), epsilon

    def update(self, arr):
        bm, bv, bc = np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / (self.count + bc)
        self.mean, self.var, self.count = new_mean, m2 / (self.count + bc), bc + self.count


class MeanStdNormalizer:
    def __init__(self, shape=(), clip=10.0, epsilon=1e-8):
        self.rms, self.clip, self.epsilon, self.read_only = RunningMeanStd(shape=shape), clip, epsilon, False

    def set_read_only(self):                # normalization.py:70-72
        self.read_only = True

    def __call__(self, x):
        x = np.asarray(x)
        if not self.read_only:
            self.rms.update(x)
        return np.clip((x - self.rms.mean) / np.sqrt(self.rms.var + self.epsilon), -self.clip, self.clip)


class RewardStdNormalizer(MeanStdNormalizer):
    def __init__(self, gamma=0.99, clip=10.0, epsilon=1e-8):
        super().__init__((), clip, epsilon)
        self.gamma, self.ret = gamma, None

    def __call__(self, x, dones):
        x = np.asarray(x)
        if not self.read_only:
            if self.ret is None:
                self.ret = np.zeros(x.shape[0])
            self.ret = self.ret * self.gamma + x
            self.rms.update(self.ret)
            self.ret[dones.astype(np.int64)] = 0
        return np.clip(x / np.sqrt(self.rms.var + self.epsilon), -self.clip, self.clip)
