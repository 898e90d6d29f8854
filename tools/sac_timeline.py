#!/usr/bin/env python3
"""In-kernel timeline of the fused SAC step's actor_grad_kernel (csrc/scg_sac.hip built with -DSCG_S_TIMING into a TAGGED scratch
library: the shipped one is untouched): shader-clock stamps of wave 0 of workgroup 0 at its phase boundaries, printed as microseconds at
2.4 GHz.   SCG_SAC_FLAGS=-DSCG_S_TIMING python tools/sac_timeline.py        (GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SCG_SAC_FLAGS', '-DSCG_S_TIMING')
NAMES = ['start', 'data-gradient operand, stored tiles and row values requested + small block filled + barrier', 'first tile: stored tiles and row values arrived',
         'loss derivatives (from the stored head: no forward pass, no tanh-Gaussian algebra)',
         'db3 + dW3 (h2 transposed)', 'dz2, dz2^T, db2, h1^T, sample cache (before the barrier)', 'barrier', 'data gradient (64 MFMA) x act\'', 'dW1 | db1 (16 MFMA) + store',
         'dW2 (64 MFMA) + 64 KB of partial stores', 'barrier', 'statistics, end']


def main():
    import torch
    from safe_control_gym_amd import _lib as L
    from safe_control_gym_amd import _sac
    so = _sac.lib_path(24, 128, 4, 'relu')
    tagged = so[:-3] + '_timing.so'
    orig = _sac.lib_path
    _sac.lib_path = lambda *a: tagged                      # (build + load the tagged library in place of the shipped one)
    if os.path.exists(tagged):
        os.remove(tagged)
    _sac.build(24, 128, 4, 'relu', force=True)
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)
    torch.manual_seed(1)
    ag = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu'), dev)
    buf = DeviceReplay(1_000_000, 24, 4, dev)
    n = 500_000
    buf.push(torch.randn(n, 24, device=dev), torch.rand(n, 4, device=dev) * 2 - 1, torch.randn(n, device=dev), torch.randn(n, 24, device=dev), torch.ones(n, device=dev))
    for _ in range(5):
        ag.update_from_buffer(buf, 4096, 8)
    torch.cuda.synchronize()
    D = _sac.lib(24, 128, 4, 'relu')
    out = (C.c_ulonglong * 32)()
    D.scg_sac_timeline.argtypes = [C.POINTER(C.c_ulonglong)]
    assert D.scg_sac_timeline(out) == 0
    t = list(out)[:16]
    print('actor_grad_kernel, wave 0 of workgroup 0, batch 4096 (one tile per workgroup), shader clock / 2.4 GHz:')
    for k in range(1, 12):
        print(f'  {NAMES[k]:100s} {(t[k] - t[k - 1]) / 2400.0:7.2f} us   (at {(t[k] - t[0]) / 2400.0:6.2f})')
    _sac.lib_path = orig
    os.remove(tagged)


if __name__ == '__main__':
    main()
