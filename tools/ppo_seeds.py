#!/usr/bin/env python3
"""PPO wall-clock-to-reward over several seeds at one batch size (bench.py's ppo leg, parametrised).

    python tools/ppo_seeds.py --envs 65536 --minibatch 262144 --seeds 6 --budget 20
Prints one JSON line: per-seed wall clock until the deterministic-policy evaluation return reaches 236 / 250 (None if the
budget ran out), iterations, best return, median."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=16384)
    ap.add_argument('--minibatch', type=int, default=None)
    ap.add_argument('--seeds', type=int, default=6)
    ap.add_argument('--budget', type=float, default=10.0)
    ap.add_argument('--lr', type=float, default=2e-3)
    ap.add_argument('--target-kl', type=float, default=0.03)
    ap.add_argument('--epochs', type=int, default=None)
    ap.add_argument('--rollout-steps', type=int, default=32)
    ap.add_argument('--mb-per-epoch', type=int, default=None, help='partial epochs: optimiser steps per epoch (PPOConfig.extra minibatches_per_epoch)')
    a = ap.parse_args()
    import torch
    import bench
    torch.cuda.set_device(0)
    res = bench.ppo_leg(torch, None, 1, 0, a.seeds, a.budget, envs=a.envs, minibatch=a.minibatch, lr=a.lr, target_kl=a.target_kl,
                        epochs=a.epochs, rollout_steps=a.rollout_steps, mb_per_epoch=a.mb_per_epoch)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
