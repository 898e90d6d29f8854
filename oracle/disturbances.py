"""Disturbances, batched.  ORACLE — test infrastructure only.

Follows envs/disturbances.py: Disturbance :6-35, DisturbanceList :38-67 (sequential
apply), ImpulseDisturbance :70-123, StepDisturbance :126-163, UniformNoise :166-192,
WhiteNoise :195-223, PeriodicNoise :233-259 (drops ``mask``; fresh random phase every
call), create_disturbance_list :285-303.

Random numbers come from ``env.draws`` (oracle/envs.py::Draws) so the same code runs
on the reference's per-env NumPy generators or on the kernels' Philox streams.
"""
import numpy as np


class _Disturbance:
    draws_at_reset = False

    def __init__(self, env, dim, mask=None, **kwargs):
        self.dim = dim
        self.mask = None
        if mask is not None:
            self.mask = np.asarray(mask, dtype=np.float32)          # :17-19
            assert self.dim == len(self.mask)

    def reset(self, env, idx, channel, item):
        pass

    def _finish(self, target, noise):
        if self.mask is not None:
            noise = noise * self.mask
        return target + noise


class ImpulseDisturbance(_Disturbance):
    def __init__(self, env, dim, mask=None, magnitude=1, step_offset=None, duration=1,
                 decay_rate=1, **kwargs):
        super().__init__(env, dim, mask)
        self.magnitude = magnitude
        self.step_offset = step_offset
        self.max_step = int(env.EPISODE_LEN_SEC / env.CTRL_TIMESTEP)      # :91
        assert duration >= 1
        assert 0 < decay_rate <= 1
        self.duration = duration
        self.decay_rate = decay_rate
        self.current_step_offset = np.zeros(env.num_envs, dtype=np.int64)
        self.current_peak_step = np.zeros(env.num_envs, dtype=np.int64)
        self.draws_at_reset = step_offset is None

    def reset(self, env, idx, channel, item):
        if self.step_offset is None:                                        # :102-106
            self.current_step_offset[idx] = env.draws.reset_integer(idx, channel, item, self.max_step)
        else:
            self.current_step_offset[idx] = self.step_offset
        self.current_peak_step[idx] = (self.current_step_offset[idx] + self.duration / 2).astype(np.int64)

    def apply(self, target, env, channel, item):
        c = env.ctrl_step_counter
        peak_offset = np.abs(c - self.current_peak_step)
        decay = np.where(peak_offset < self.duration / 2,
                         np.power(float(self.decay_rate), peak_offset.astype(np.float64)), 0.0)
        noise = np.where(c >= self.current_step_offset, self.magnitude * decay, 0.0)   # :112-119
        return self._finish(target, noise[:, None])


class StepDisturbance(_Disturbance):
    def __init__(self, env, dim, mask=None, magnitude=1, step_offset=None, **kwargs):
        super().__init__(env, dim, mask)
        self.magnitude = magnitude
        self.step_offset = step_offset
        self.max_step = int(env.EPISODE_LEN_SEC / env.CTRL_TIMESTEP)
        self.current_step_offset = np.zeros(env.num_envs, dtype=np.int64)
        self.draws_at_reset = step_offset is None

    def reset(self, env, idx, channel, item):
        if self.step_offset is None:                                        # :148-151
            self.current_step_offset[idx] = env.draws.reset_integer(idx, channel, item, self.max_step)
        else:
            self.current_step_offset[idx] = self.step_offset

    def apply(self, target, env, channel, item):
        active = env.ctrl_step_counter >= self.current_step_offset
        if self.mask is not None:
            # :156-160 — ``noise`` is a Python scalar here, so ``noise *= mask`` (float32 array) is
            # evaluated in float32 under NumPy >= 2 promotion rules (reference pins numpy ^2.2).
            noise = np.where(active[:, None], np.float32(self.magnitude) * self.mask, np.float32(0.0))
            return target + noise.astype(np.float64)
        noise = np.where(active, float(self.magnitude), 0.0)
        return target + noise[:, None]


class UniformNoise(_Disturbance):
    def __init__(self, env, dim, mask=None, low=0.0, high=1.0, **kwargs):
        super().__init__(env, dim, mask)
        if isinstance(low, float):                                          # :173-178
            self.low = np.asarray([low] * self.dim)
        elif isinstance(low, list):
            self.low = np.asarray(low)
        else:
            raise ValueError('[ERROR] UniformNoise.__init__(): low must be specified as a float or list.')
        if isinstance(high, float):                                         # :180-185 (sic: tests ``low``)
            self.high = np.asarray([high] * self.dim)
        elif isinstance(low, list):
            self.high = np.asarray(high)
        else:
            raise ValueError('[ERROR] UniformNoise.__init__(): high must be specified as a float or list.')

    def apply(self, target, env, channel, item):
        noise = env.draws.step_uniform(channel, item, self.low, self.high)   # :188
        return self._finish(target, noise)


class WhiteNoise(_Disturbance):
    def __init__(self, env, dim, mask=None, std=1.0, **kwargs):
        super().__init__(env, dim, mask)
        if isinstance(std, float):                                          # :207-212
            self.std = np.asarray([std] * self.dim)
        elif isinstance(std, list):
            self.std = np.asarray(std)
        else:
            raise ValueError('[ERROR] WhiteNoise.__init__(): std must be specified as a float or list.')
        assert self.dim == len(self.std), 'std shape should be the same as dim.'

    def apply(self, target, env, channel, item):
        noise = env.draws.step_normal(channel, item, self.std)               # :219
        return self._finish(target, noise)


class PeriodicNoise(_Disturbance):
    def __init__(self, env, dim, mask=None, scale=1.0, frequency=1.0, **kwargs):
        super().__init__(env, dim)          # :244 — mask is dropped upstream
        self.scale = scale
        self.frequency = frequency

    def apply(self, target, env, channel, item):
        lo = np.full(self.dim, -np.pi)
        hi = np.full(self.dim, np.pi)
        phase = env.draws.step_uniform(channel, item, lo, hi)                # :253
        t = env.pyb_step_counter * env.PYB_TIMESTEP                          # :254
        noise = self.scale * np.sin(2 * np.pi * self.frequency * t[:, None] + phase)
        return self._finish(target, noise)


DISTURBANCE_TYPES = {'impulse': ImpulseDisturbance, 'step': StepDisturbance,
                     'uniform': UniformNoise, 'white_noise': WhiteNoise, 'periodic': PeriodicNoise}


class DisturbanceList:
    def __init__(self, disturbances, channel):
        self.disturbances = disturbances
        self.channel = channel

    def reset(self, env, idx):
        for k, d in enumerate(self.disturbances):
            d.reset(env, idx, self.channel, k)

    def apply(self, target, env):
        out = target
        for k, d in enumerate(self.disturbances):
            out = d.apply(out, env, self.channel, k)
        return out


def create_disturbance_list(specs, shared_args, env, channel):
    """disturbances.py:285-303."""
    out = []
    for spec in specs:
        assert 'disturbance_func' in spec
        cls = DISTURBANCE_TYPES[spec['disturbance_func']]
        cfg = {k: v for k, v in spec.items() if k != 'disturbance_func'}
        out.append(cls(env, **shared_args, **cfg))
    return DisturbanceList(out, channel)
