#!/usr/bin/env python3
"""SAC on the HIP rollout engine (BASELINE.json config #5 shape by default: Quadrotor3D figure-8 tracking with randomised
inertial properties, white-noise dynamics disturbance and constraint evaluation).

    python examples/train_sac.py --envs 4096 --max-seconds 30
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_sac.py     # env shards + RCCL

Prints one JSON line per log interval and a summary line (env-steps/s incl. learning, mean return of finished episodes).
"""
import argparse, json, os, sys, time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_control_gym_amd import parallel                                   # noqa: E402
from safe_control_gym_amd.registration import load_task                     # noqa: E402
from safe_control_gym_amd.sac import SAC, SACConfig                         # noqa: E402
from safe_control_gym_amd.vec_env import HipVecEnv                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--task', default='quadrotor_3D_track_disturbed')
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--hidden', type=int, default=128)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--updates-per-step', type=int, default=8)
    ap.add_argument('--lr', type=float, default=3e-4)
    ap.add_argument('--buffer', type=int, default=2_000_000)
    ap.add_argument('--warm-up-steps', type=int, default=65536)
    ap.add_argument('--max-seconds', type=float, default=30.0)
    ap.add_argument('--max-env-steps', type=float, default=1e8)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--save-final', default=None, help='write <path>.rank<r>.pt: final flat parameters of this rank (tests: ranks stay in lock-step)')
    args = ap.parse_args()
    rank, world = parallel.init_distributed()
    env_id, cfg = load_task(args.task)
    env = HipVecEnv(env_id, args.envs, seed=args.seed, env_id_offset=rank * args.envs, return_numpy=False, **cfg)
    scfg = SACConfig(hidden_dim=args.hidden, train_batch_size=args.batch, actor_lr=args.lr, critic_lr=args.lr,
                     warm_up_steps=args.warm_up_steps, train_interval=args.envs * world, max_buffer_size=args.buffer,
                     extra={'updates_per_step': args.updates_per_step})
    sac = SAC(env, scfg, seed=args.seed)
    flat = lambda: torch.cat([p.detach().reshape(-1) for p in sac.agent.ac.parameters()]).cpu()      # noqa: E731
    init_params = flat()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it, ep_n, ep_ret = 0, 0.0, 0.0
    last_log = t0
    while True:
        # the gradient exchange sits inside train_step: every rank must leave the loop on the same iteration — rank 0 decides
        stop = torch.tensor([1.0 if (sac.total_steps >= args.max_env_steps or time.perf_counter() - t0 >= args.max_seconds) else 0.0],
                            device=env.device)
        parallel.broadcast_(stop, 0)
        if stop.item() > 0:
            break
        res = sac.train_step()
        it += 1
        d = env.out.done.to(torch.float32)
        ep_n += float(d.sum()) if it % 16 == 0 else 0.0                       # sampled: a host sync every 16th step only
        ep_ret += float((env.out.fin_return * d).sum()) if it % 16 == 0 else 0.0
        if rank == 0 and not args.quiet and time.perf_counter() - last_log > 5.0:
            last_log = time.perf_counter()
            print(json.dumps({'step': sac.total_steps, 'wall_clock': last_log - t0, 'mean_episode_return': ep_ret / max(ep_n, 1.0),
                              **{k: v for k, v in res.items() if k in ('policy_loss', 'critic_loss', 'updates')}}), flush=True)
            ep_n, ep_ret = 0.0, 0.0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if args.save_final:
        torch.save({'params': flat(), 'init_params': init_params, 'env_id_offset': env.env_id_offset, 'vector_steps': it,
                    'fused_update': bool(sac.agent.use_fused), 'first_obs': sac.obs[:8].cpu()}, f'{args.save_final}.rank{rank}.pt')
    if rank == 0:
        print(json.dumps({'summary': True, 'task': args.task, 'n_gpus': world, 'envs_per_gpu': args.envs, 'vector_steps': it,
                          'env_steps': sac.total_steps, 'wall_clock_s': wall, 'env_steps_per_s_incl_learning': sac.total_steps / wall,
                          'hyper': vars(args)}))
    env.close()


if __name__ == '__main__':
    main()
