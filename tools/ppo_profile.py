#!/usr/bin/env python3
"""Per-phase wall clock of PPO iterations on the HIP engine: rollout front end, update, evaluation.
usage: ppo_profile.py [--envs N] [--iters K] [--no-fused] [--minibatch B] [--epochs E]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=16384)
    ap.add_argument('--rollout-steps', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=4)
    ap.add_argument('--minibatch', type=int, default=65536)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--mb-per-epoch', type=int, default=None, help='partial epochs (PPOConfig.extra minibatches_per_epoch)')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--fused-rollout', action='store_true')
    args = ap.parse_args()
    import torch
    from safe_control_gym_amd.ppo import PPO, PPOConfig, evaluate
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    torch.cuda.set_device(0)
    env_id, cfg = load_task('quadrotor_2D_track')
    pol = (128, 'tanh') if args.fused_rollout else None
    env = HipVecEnv(env_id, args.envs, seed=2, return_numpy=False, policy=pol, **cfg)
    eval_env = HipVecEnv(env_id, 256, seed=222, return_numpy=False, policy=pol, **dict(cfg, randomized_init=False))
    pcfg = PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, target_kl=0.03, opt_epochs=args.epochs,
                     mini_batch_size=args.minibatch, actor_lr=2e-3, critic_lr=2e-3, rollout_batch_size=args.envs,
                     rollout_steps=args.rollout_steps, extra={'fused_update': not args.no_fused, 'fused_rollout': args.fused_rollout, 'iteration_graph': False,   # (per-phase clocks need the phases as separate enqueues)
                            **({'minibatches_per_epoch': args.mb_per_epoch} if args.mb_per_epoch else {})})
    ppo = PPO(env, pcfg, seed=2)
    tot = {'collect': 0.0, 'update': 0.0, 'eval': 0.0}
    for it in range(args.iters + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ppo.train_step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ev = evaluate(ppo.agent.ac, eval_env, policy=ppo._policy_struct(True) if pol else None)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 3:
            tot['collect'] += res['collect_time']; tot['update'] += (t1 - t0) - res['collect_time']; tot['eval'] += t2 - t1
    out = {k: round(1e3 * v / args.iters, 3) for k, v in tot.items()}
    out.update(fused=not args.no_fused, envs=args.envs, T=args.rollout_steps, minibatch=args.minibatch, epochs=args.epochs,
               eval_return=ev['ep_return'])
    print(json.dumps(out))


if __name__ == '__main__':
    main()
