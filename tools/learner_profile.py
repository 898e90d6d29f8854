#!/usr/bin/env python3
"""The learning legs of bench.py as plain loops for rocprofv3 --kernel-trace --stats (tools/profile_round6.sh): K iterations of the PPO
leg's configuration (65 536 envs x 32 steps, 3 partial epochs x 16 minibatches of 16 256 — one HIP-graph replay per iteration), or K
learning vector steps of the SAC leg's (2048 envs, batch 4096, 16 gradient steps per vector step).  Prints one JSON line with the
number of iterations that EXECUTED on the device (what the kernel totals of the trace are divided by) and the wall clock per iteration
of the untraced-equivalent loop (host never waits: lazy train_step, bounded run-ahead), so that the same command gives both halves of
bench.py's `iteration_ms` when it is run with and without the tracer.

    python tools/learner_profile.py ppo [--iters 40] [--envs 65536] [--no-graph]
    python tools/learner_profile.py sac [--iters 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_ppo(args, torch):
    import bench
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task('quadrotor_2D_track')
    env = HipVecEnv(env_id, args.envs, seed=1, return_numpy=False, policy=(128, 'tanh'), **cfg)
    extra = {'minibatches_per_epoch': args.mb_per_epoch} if args.mb_per_epoch else {}
    if args.no_graph:
        extra['iteration_graph'] = False
    pcfg = PPOConfig(hidden_dim=128, activation='tanh', gamma=0.99, use_gae=True, gae_lambda=0.95, target_kl=0.03, entropy_coef=0.01,
                     opt_epochs=args.epochs, mini_batch_size=args.minibatch, actor_lr=2e-3, critic_lr=2e-3, rollout_batch_size=args.envs,
                     rollout_steps=32, extra=extra)
    ppo = PPO(env, pcfg, seed=1)
    aev = None
    if args.eval_chunk is not None:                     # bench.py's asynchronous evaluation beside the loop (--eval-chunk 0 = one launch)
        from safe_control_gym_amd.ppo import AsyncEvaluator
        ev_cfg = bench.eval_task_config(cfg, bench.EVAL_INIT_RAND_Q2)
        eval_env = HipVecEnv(env_id, bench.EVAL_ENVS, seed=111, return_numpy=False, policy=(128, 'tanh'), **ev_cfg)
        aev = AsyncEvaluator(ppo, eval_env)
        aev.chunk = args.eval_chunk or None
    for _ in range(3):                                  # eager, captured (+ replayed), replayed
        ppo.train_step(lazy=True)
        if aev:
            aev.launch()
    torch.cuda.synchronize()
    if aev:
        aev.poll(wait=True)
    pending, ends = [], []
    t0 = time.perf_counter()
    for _ in range(args.iters):
        res = ppo.train_step(lazy=True)
        if aev:
            aev.launch()
            aev.poll()
        ends.append(res['events'])
        pending.append(res['events'][2])
        while len(pending) > 2:
            pending.pop(0).synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gaps = sorted(a[2].elapsed_time(b[2]) for a, b in zip(ends[:-1], ends[1:]))
    n_steps = args.epochs * min(args.envs * 32 // args.minibatch, args.mb_per_epoch or 10 ** 9)
    stats = ppo.agent.stats_of(res['stats_dev'], res['minibatches'])
    env.close()
    return {'mode': 'ppo', 'key': f'ppo/{args.envs}/{n_steps}x{args.minibatch}', 'iterations_executed': 3 + args.iters, 'iterations_timed': args.iters,
            'wall_ms_per_iteration': 1e3 * wall / args.iters, 'device_ms_per_iteration_median': gaps[len(gaps) // 2],
            'flops_per_iteration': (args.envs * 32 * bench.mlp_flops(12, 128, 2) + args.envs * 33 * bench.mlp_flops(12, 128, 1)
                                    + n_steps * args.minibatch * (bench.mlp_flops(12, 128, 2, True) + bench.mlp_flops(12, 128, 1, True))),
            'optimiser_steps_per_iteration': n_steps, 'host_path': 'per-launch enqueue' if args.no_graph else 'one HIP-graph replay per iteration',
            'last_update': stats}


def run_sac(args, torch):
    import bench
    from safe_control_gym_amd.sac import SAC, SACConfig
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg, _ = bench.sac_task_config(None)
    envs, batch, ups = 2048, 4096, 16
    env = HipVecEnv(env_id, envs, seed=1, return_numpy=False, **cfg)
    warm = 8 * envs
    sac = SAC(env, SACConfig(hidden_dim=128, activation='relu', train_batch_size=batch, actor_lr=1e-3, critic_lr=1e-3, warm_up_steps=warm,
                             train_interval=envs, max_buffer_size=1_000_000, extra={'updates_per_step': ups}), seed=1)
    n_warm = 0
    while sac.total_steps <= warm:                      # uniform-action warm-up: collector only
        sac.train_step(lazy=True)
        n_warm += 1
    for _ in range(4):                                  # first learning steps (graph captures)
        sac.train_step(lazy=True)
    torch.cuda.synchronize()
    pending = []
    t0 = time.perf_counter()
    for _ in range(args.iters):
        sac.train_step(lazy=True)
        e = torch.cuda.Event()
        e.record()
        pending.append(e)
        while len(pending) > 4:
            pending.pop(0).synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    spec = env.spec
    a_f, q_f = bench.mlp_flops(spec.obs_dim, 128, 2 * spec.nu), bench.mlp_flops(spec.obs_dim + spec.nu, 128, 1)
    env.close()
    return {'mode': 'sac', 'key': f'sac/{batch}/{ups}', 'vector_steps_executed': n_warm + 4 + args.iters, 'learning_vector_steps_executed': 4 + args.iters,
            'gradient_steps_executed': ups * (4 + args.iters), 'wall_ms_per_vector_step': 1e3 * wall / args.iters,
            'flops_per_gradient_step': batch * (2 * a_f + 4 * q_f + 2 * a_f + 2 * q_f + 6 * q_f)}       # (bench.py: actor forward counted once)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('mode', choices=['ppo', 'sac'])
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--minibatch', type=int, default=16256)
    ap.add_argument('--mb-per-epoch', type=int, default=16)
    ap.add_argument('--eval-chunk', type=int, default=None, help='run bench.py\'s asynchronous evaluation beside the loop, this many control steps per launch (0 = one launch)')
    ap.add_argument('--no-graph', action='store_true', help='per-launch enqueue instead of the iteration graph (same launches)')
    args = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    out = run_ppo(args, torch) if args.mode == 'ppo' else run_sac(args, torch)
    print('LEARNER_PROFILE ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
