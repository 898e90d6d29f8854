"""Device-side rollout utilities: GAE/returns through scg_gae.

Replaces compute_returns_and_advantages
(/root/reference/safe_control_gym/controllers/ppo/ppo_utils.py:374-400) for [T, N] device buffers.
"""
import ctypes as C

import torch

from safe_control_gym_amd import _lib as L


def gae_returns(rew, v, mask, terminal_v, last_v, gamma=0.99, gae_lambda=0.95, use_gae=True, out=None):
    """rew, v, mask, terminal_v: [T, N] contiguous device tensors; last_v: [N].

    Like the reference, ``rew`` is updated in place with ``gamma * terminal_v`` (time-limit bootstrap,
    ppo_utils.py:389).  Returns (ret, adv), both [T, N]; ``out=(ret, adv)`` writes into caller-owned buffers (a collector
    that calls this every iteration keeps them: no allocator traffic between the kernels)."""
    if not rew.is_cuda:
        raise L.ScgError('gae_returns needs device tensors; there is no CPU fallback')
    T, N = rew.shape
    dt = rew.dtype
    if dt not in (torch.float32, torch.float64):
        raise ValueError('float32 / float64 only')
    for t in (v, mask, last_v) + ((terminal_v,) if terminal_v is not None else ()):
        if t.dtype != dt or not t.is_contiguous() or t.device != rew.device:
            raise ValueError('all GAE buffers must be contiguous, same dtype and device')
    if not rew.is_contiguous():
        raise ValueError('rew must be contiguous')
    if out is None:
        ret, adv = torch.empty_like(rew), torch.empty_like(rew)
    else:
        ret, adv = out
        for t in (ret, adv):
            if t.shape != rew.shape or t.dtype != dt or not t.is_contiguous() or t.device != rew.device:
                raise ValueError('out buffers must match rew (shape, dtype, device, contiguous)')
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    with torch.cuda.device(rew.device):
        stream = C.c_void_p(torch.cuda.current_stream(rew.device).cuda_stream)
        L.check(L.lib().scg_gae(L.F64 if dt == torch.float64 else L.F32, p(rew), p(v), p(mask), p(terminal_v),
                                p(last_v), p(ret), p(adv), int(T), int(N), float(gamma), float(gae_lambda),
                                int(bool(use_gae)), stream))
    return ret, adv
