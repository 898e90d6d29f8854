"""Observation / reward normalisers of the RL collectors, on device tensors.

Mirrors /root/reference/safe_control_gym/math_and_models/normalization.py:
  RunningMeanStd        :13-55     parallel-variance update from batch moments (count starts at epsilon = 1e-4)
  MeanStdNormalizer     :88-124    clip((x - mean) / sqrt(var + eps), -clip, clip), updates unless read_only
  RewardStdNormalizer   :127-159   running discounted return per env, reward scaled by its std (no centring)
The statistics are float64 like upstream's; inputs/outputs keep the collector's dtype.  With several ranks the batch
moments are summed over ranks first (one tiny all-reduce), so every rank holds the same statistics.

Upstream quirk kept by default (`faithful_reset=True`): RewardStdNormalizer resets the running returns with
`self.ret[dones.astype(np.long)] = 0` (:158) — an INDEX array of zeros and ones, i.e. env 0 is reset whenever any env is
not done and env 1 whenever any env is done, the other envs never.  `faithful_reset=False` resets exactly the done envs.
"""
import torch

from safe_control_gym_amd import parallel


class RunningMeanStd:
    def __init__(self, shape=(), device='cpu', epsilon=1e-4):
        self.mean = torch.zeros(shape, dtype=torch.float64, device=device)
        self.var = torch.ones(shape, dtype=torch.float64, device=device)
        self.count = torch.full((), float(epsilon), dtype=torch.float64, device=device)

    def update(self, arr):
        """arr: [batch, *shape]."""
        x = arr.to(torch.float64)
        n = torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)
        if parallel.world_size() > 1:
            flat = torch.cat([n, x.sum(0).reshape(-1), (x * x).sum(0).reshape(-1)])
            parallel.all_reduce_sum_(flat)
            k = self.mean.numel()
            n, s1, s2 = flat[:1], flat[1:1 + k].reshape(self.mean.shape), flat[1 + k:].reshape(self.mean.shape)
            batch_mean = s1 / n
            batch_var = torch.clamp(s2 / n - batch_mean * batch_mean, min=0.0)
        else:
            batch_mean = x.mean(0)
            batch_var = x.var(0, unbiased=False)
        self.update_from_moments(batch_mean, batch_var, n.reshape(()))

    def update_masked(self, arr, mask):
        """Update from the rows of `arr` selected by the boolean `mask` — the statistics of `update(arr[mask])`, computed with static
        shapes and no host value (a data-dependent row count inside a captured graph); no selected row: the statistics stay."""
        x = arr.to(torch.float64)
        w = mask.to(torch.float64).reshape((-1,) + (1,) * (x.dim() - 1))
        n, s1 = w.sum().reshape(1), (w * x).sum(0)
        if parallel.world_size() > 1:
            flat = torch.cat([n, s1.reshape(-1)])
            parallel.all_reduce_sum_(flat)
            n, s1 = flat[:1], flat[1:].reshape(s1.shape)
        n1 = n.clamp(min=1.0)
        batch_mean = s1 / n1
        s2 = (w * (x - batch_mean) ** 2).sum(0)
        if parallel.world_size() > 1:
            parallel.all_reduce_sum_(s2)
        self.update_from_moments(batch_mean, s2 / n1, n.reshape(()))

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + delta * delta * self.count * batch_count / tot
        self.mean.copy_(new_mean)
        self.var.copy_(m2 / tot)
        self.count.copy_(tot)


class BaseNormalizer:
    """Identity (normalization.py:58-85)."""

    def __init__(self, read_only=False):
        self.read_only = read_only

    def set_read_only(self):
        self.read_only = True

    def unset_read_only(self):
        self.read_only = False

    def __call__(self, x, *args, **kwargs):
        return x

    def state_dict(self):
        return {}

    def load_state_dict(self, _):
        pass


class MeanStdNormalizer(BaseNormalizer):
    def __init__(self, shape=(), device='cpu', read_only=False, clip=10.0, epsilon=1e-8):
        super().__init__(read_only)
        self.rms = RunningMeanStd(shape, device)
        self.clip, self.epsilon = float(clip), float(epsilon)

    def __call__(self, x, mask=None):
        """`mask` (bool [batch]): only those rows enter the statistics — upstream's call on a gathered subset of rows
        (sac.py:296-297: the truncated envs' terminal observations); every row is normalised."""
        if not self.read_only:
            if mask is None:
                self.rms.update(x)
            else:
                self.rms.update_masked(x, mask)
        y = (x.to(torch.float64) - self.rms.mean) / torch.sqrt(self.rms.var + self.epsilon)
        return y.clamp(-self.clip, self.clip).to(x.dtype)

    def state_dict(self):
        return {'mean': self.rms.mean.cpu().numpy(), 'var': self.rms.var.cpu().numpy()}

    def load_state_dict(self, saved):
        self.rms.mean.copy_(torch.as_tensor(saved['mean'], dtype=torch.float64))
        self.rms.var.copy_(torch.as_tensor(saved['var'], dtype=torch.float64))


class RewardStdNormalizer(MeanStdNormalizer):
    def __init__(self, gamma=0.99, device='cpu', read_only=False, clip=10.0, epsilon=1e-8, faithful_reset=True):
        super().__init__((), device, read_only, clip, epsilon)
        self.gamma = float(gamma)
        self.ret = None
        self.faithful_reset = faithful_reset

    def __call__(self, x, dones):
        if not self.read_only:
            if self.ret is None:
                self.ret = torch.zeros(x.shape[0], dtype=torch.float64, device=x.device)
            self.ret.mul_(self.gamma).add_(x.to(torch.float64))
            self.rms.update(self.ret)
            d = dones.to(torch.bool)
            if self.faithful_reset:
                # ret[dones.astype(long)] = 0: index 0 for every not-done env, index 1 for every done env
                any_not, any_done = (~d).any(), d.any()
                self.ret[0] = torch.where(any_not, torch.zeros_like(self.ret[0]), self.ret[0])
                if self.ret.shape[0] > 1:
                    self.ret[1] = torch.where(any_done, torch.zeros_like(self.ret[1]), self.ret[1])
            else:
                self.ret.masked_fill_(d, 0.0)
        y = x.to(torch.float64) / torch.sqrt(self.rms.var + self.epsilon)
        return y.clamp(-self.clip, self.clip).to(x.dtype)
