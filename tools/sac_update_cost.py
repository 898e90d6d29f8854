#!/usr/bin/env python3
"""ms per SAC gradient step at the production shape (24-128-128, 4 actions): fused (scg_sac_update, graph of n steps) vs the
captured graph of PyTorch kernels.   python tools/sac_update_cost.py [--batch 4096] [--steps 8]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--reps', type=int, default=50)
    a = ap.parse_args()
    import torch
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)
    out = {'batch': a.batch, 'steps_per_call': a.steps}
    for tag, extra in (('fused', {}), ('torch_graph', {'fused_update': False})):
        torch.manual_seed(1)
        ag = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu', extra=extra), dev)
        buf = DeviceReplay(1_000_000, 24, 4, dev)
        n = 500_000
        buf.push(torch.randn(n, 24, device=dev), torch.rand(n, 4, device=dev) * 2 - 1, torch.randn(n, device=dev), torch.randn(n, 24, device=dev),
                 torch.ones(n, device=dev))
        for _ in range(3):
            ag.update_from_buffer(buf, a.batch, a.steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            ag.update_from_buffer(buf, a.batch, a.steps)
        torch.cuda.synchronize()
        out[tag + '_ms_per_gradient_step'] = 1e3 * (time.perf_counter() - t0) / (a.reps * a.steps)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
