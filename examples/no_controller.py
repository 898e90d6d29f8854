#!/usr/bin/env python3
"""Open-loop random-action throughput, the measurement behind the reference's README "Performance" table
(/root/reference/examples/no_controller/verbose_api.py:111-116: steps/s and speed-up = simulated time / wall-clock time),
on the HIP env: one env through the reference-compatible facade, and a batch through the tensor API.

    python examples/no_controller.py --task quadrotor_2D_track --envs 65536
"""
import argparse, os, sys, time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_control_gym_amd.registration import load_task, make                 # noqa: E402
from safe_control_gym_amd.vec_env import HipVecEnv                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--task', default='quadrotor_2D_track')
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--seconds', type=float, default=3.0)
    args = ap.parse_args()
    env_id, cfg = load_task(args.task)
    # --- one env, reference API (numpy in / out, info dicts): dominated by the per-call host overhead
    env = make(env_id, **cfg)
    obs, info = env.reset(seed=1)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        obs, rew, done, info = env.step(env.action_space.sample())
        n += 1
        if done:
            env.reset()
    el = time.perf_counter() - t0
    print(f'single env (facade): {n / el:10.0f} env-steps/s  = {n / el / env.CTRL_FREQ:8.1f} x real time '
          f'(reference README, PyBullet on a laptop CPU: 381-464 quadrotor / 1120-1236 cartpole env-steps/s)')
    env.close()
    # --- batch, tensor API, one kernel launch per control step, auto-reset inside the kernel
    venv = HipVecEnv(env_id, args.envs, seed=1, return_numpy=False, **cfg)
    venv.reset_tensors()
    acts = [torch.rand(args.envs, venv.spec.nu, device=venv.device) * 2 - 1 for _ in range(16)]
    for k in range(50):
        venv.step_tensors(acts[k % 16])
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        for k in range(200):
            venv.step_tensors(acts[k % 16])
        torch.cuda.synchronize()
        n += 200
    el = time.perf_counter() - t0
    rate = n * args.envs / el
    print(f'{args.envs} envs (step_tensors, Python launch loop): {rate:.3e} env-steps/s = {rate / venv.spec.CTRL_FREQ:.3e} x real time '
          f'(bench.py replays the same launches from a HIP graph)')
    r, d, v, _ = venv.rollout_random(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    venv.rollout_random(1000)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f'{args.envs} envs (scg_rollout_random, 1000 fused steps per launch): {1000 * args.envs / el:.3e} env-steps/s')
    venv.close()


if __name__ == '__main__':
    main()
