#!/usr/bin/env python3
"""Per-iteration wall time of bench.py's PPO leg configuration (65 536 envs, 2 partial epochs x 32 x 16 256): the first iterations against the
steady state — how much of the 0.66 s wall clock to reward is one-time start-up (module loads, first launches) inside the clock."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from safe_control_gym_amd.ppo import PPO, PPOConfig  # noqa: E402
from safe_control_gym_amd.registration import load_task  # noqa: E402
from safe_control_gym_amd.vec_env import HipVecEnv  # noqa: E402

torch.cuda.set_device(0)
env_id, cfg = load_task('quadrotor_2D_track')
for seed in (1, 2):
    env = HipVecEnv(env_id, 65536, seed=seed, return_numpy=False, policy=(128, 'tanh'), **cfg)
    ppo = PPO(env, PPOConfig(hidden_dim=128, activation='tanh', gamma=0.99, use_gae=True, gae_lambda=0.95, target_kl=0.03, entropy_coef=0.01,
                             opt_epochs=2, mini_batch_size=16256, actor_lr=2e-3, critic_lr=2e-3, rollout_batch_size=65536, rollout_steps=32,
                             extra={'minibatches_per_epoch': 32, 'fused_step': '--no-fused-step' not in sys.argv}), seed=seed)
    torch.cuda.synchronize()
    ts = []
    for it in range(40):
        t0 = time.perf_counter()
        ppo.train_step()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print('fused_step' if ppo.agent._fused_step_ok else 'grad + adam', f'seed {seed}: first iterations ms', [round(t, 2) for t in ts[:6]], 'steady state ms (mean of 20..39)', round(sum(ts[20:]) / 20, 3),
          'start-up excess ms', round(sum(ts[:6]) - 6 * sum(ts[20:]) / 20, 1))
    env.close()
