#!/usr/bin/env python3
"""Fixed vs per-tile cost of one scg_ppo_grad call (gradient kernel + partial-sum reduction): time it for minibatches of
16 384 ... 262 144 rows (1 ... 16 column tiles per wave at 128 workgroups per network) and fit T = F + c * tiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.test_gpu_learn import _agent, _data  # noqa: E402

ag = _agent(12, 128, 2, 'tanh')
M = 524288
data = _data(12, 2, M, ag)
rows = []
for mb in (16384, 32768, 65536, 131072, 262144):
    F = ag._build_fused(data, mb)
    F['idx'].copy_(torch.randperm(M, device='cuda')[:mb].to(torch.int32))
    for _ in range(3):
        ag._fused_grad(F)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        ag._fused_grad(F)
    ev[1].record()
    torch.cuda.synchronize()
    us = 1e3 * ev[0].elapsed_time(ev[1]) / 20
    tiles = mb // 32 // (F['args'].n_workgroups * 4)
    rows.append((mb, F['args'].n_workgroups, tiles, us))
    print(f'minibatch {mb:7d}  workgroups/net {F["args"].n_workgroups:4d}  tiles/wave {tiles:3d}  {us:8.1f} us per call')
(m0, _, t0, u0), (m1, _, t1, u1) = rows[2], rows[4]
c = (u1 - u0) / (t1 - t0)
print(f'per tile {c:.1f} us, fixed {u0 - c * t0:.1f} us (from the 65 536 and 262 144 rows)')
# the Adam kernel
F = ag._build_fused(data, 65536)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize(); ev[0].record()
for _ in range(50):
    ag._fused_adam(F)
ev[1].record(); torch.cuda.synchronize()
print(f'adam_gated: {1e3 * ev[0].elapsed_time(ev[1]) / 50:.1f} us per call')

if '--timeline' in sys.argv:            # needs a library built with SCG_LEARN_FLAGS=-DSCG_L_TIMING; --mb 16256: the one-tile form (round 6)
    mbt = int(sys.argv[sys.argv.index('--mb') + 1]) if '--mb' in sys.argv else 65536
    F = ag._build_fused(data, mbt)
    F['idx'].copy_(torch.randperm(M, device='cuda')[:mbt].to(torch.int32))
    print(f'timeline at minibatch {mbt}: {F["args"].n_workgroups} workgroups per network, {mbt // 32 / (F["args"].n_workgroups * 4):.2f} tiles per wave')
    for _ in range(3):
        ag._fused_grad(F)
    torch.cuda.synchronize()
    T = F['ws'][-256:].view(torch.int64).cpu().tolist()
    t = T[:6]
    names = ['fill + barrier', 'tile loop', 'small-gradient sums + barrier', 'dW2 staging (one-tile form: products + cross-wave sum + row stores)', 'partial vector write']
    for n, a, b in zip(names, t, t[1:]):
        print(f'  {n:75s} {(b - a) / 2400.0:8.2f} us')
    # one warm tile (the wave's last), phase boundaries; the two forward-pass stamps sit between 'inputs' and 'loss derivatives'
    tt = [T[8], T[9], T[24], T[25]] + T[10:16]
    names = ['index -> observation gather, sample cache', 'forward layer 1 (+ activations)', 'forward layer 2 (+ activations)', 'output layer',
             'loss derivatives, db3', 'dW3 (4 transposes + vector unit)', 'dz2, h1 transposes', 'data gradient + dW1 | db1', 'dW2 + db2 (4 transposes)']
    for n, a, b in zip(names, tt, tt[1:]):
        print(f'    tile: {n:44s} {(b - a) / 2400.0:8.2f} us')
    print(f'    tile total {(tt[-1] - tt[0]) / 2400.0:8.2f} us   (shader clock taken as 2.4 GHz)')
