#!/usr/bin/env python3
"""The K-steps-per-launch kernels under rocprofv3 (tools/profile_round4.sh): direct launches, no HIP graph, one mode per process.

    python tools/seq_profile.py --mode sequence_all|sequence_collector|rollout_policy [--envs 65536] [--reps 40] [--task quadrotor_2D_track]

sequence_all        scg_step_sequence, K = 8, every output of scg_step stacked [K] (bench.py: sequence.all_outputs)
sequence_collector  scg_step_sequence, K = 32, what a rollout collector keeps (obs, reward, done, flags, terminal obs)
rollout_policy      scg_rollout_policy through PPO._collect_fused: 32 control steps with the 12-128-128-2 tanh actor in the kernel
                    (+ the two batched critic passes, scg_gae, the advantage moments — their kernels show up separately)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', required=True, choices=['sequence_all', 'sequence_collector', 'rollout_policy'])
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--task', default='quadrotor_2D_track')
    a = ap.parse_args()
    import torch
    import bench
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    torch.cuda.set_device(0)
    env_id, cfg = load_task(a.task)
    if a.mode == 'rollout_policy':
        from safe_control_gym_amd.ppo import PPO, PPOConfig
        env = HipVecEnv(env_id, a.envs, seed=7, return_numpy=False, policy=(128, 'tanh'), **cfg)
        ppo = PPO(env, PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, rollout_batch_size=a.envs, rollout_steps=32, mini_batch_size=65536), seed=7)
        assert ppo._fused_rollout
        for _ in range(a.reps):
            ppo._collect_fused()
        torch.cuda.synchronize()
        print('rollout_policy', a.envs, 'envs x 32 steps x', a.reps)
        return
    env = HipVecEnv(env_id, a.envs, seed=7, return_numpy=False, **cfg)
    K, kw = (bench.SEQ_K_ALL, dict(terminal_obs=True, mse=True, c_values=True, fin_stats=True, state=True, noisy_action=True)) \
        if a.mode == 'sequence_all' else (bench.SEQ_K, dict(terminal_obs=True))
    acts = torch.rand(K, a.envs, env.spec.nu, device=env.device) * 2 - 1
    env.reset_tensors()
    out = env.step_sequence(acts, **kw)
    for _ in range(a.reps):
        env.step_sequence(acts, out=out)
    torch.cuda.synchronize()
    print(a.mode, a.envs, 'envs K', K, 'x', a.reps)


if __name__ == '__main__':
    main()
