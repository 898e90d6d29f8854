#!/usr/bin/env python3
"""Wall-clock-to-reward sweep of examples/train_ppo.py (run on the GPU box).  usage: ppo_sweep.py SECONDS CFG [CFG...]
CFG = envs:T:epochs:minibatch:lr:target_kl[:hidden]"""
import json, subprocess, sys
budget = sys.argv[1]
for cfgs in sys.argv[2:]:
    f = cfgs.split(':')
    for seed in (2, 3):
        cmd = ['python', 'examples/train_ppo.py', '--envs', f[0], '--rollout-steps', f[1], '--epochs', f[2], '--minibatch', f[3],
               '--lr', f[4], '--target-kl', f[5], '--hidden', f[6] if len(f) > 6 else '128', '--target-return', '236',
               '--eval-every', '4', '--seed', str(seed), '--quiet', '--max-seconds', budget, '--max-env-steps', '2e9']
        out = subprocess.run(cmd, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.strip().split('\n')[-1]
        try:
            d = json.loads(out)
            print(cfgs, 'seed', seed, 'iters', d['iterations'], 'best %.1f' % d['best_eval_return'], 'to_target', d['wall_clock_to_target_s'] and round(d['wall_clock_to_target_s'], 2),
                  'env-steps %.2e' % d['env_steps'], flush=True)
        except Exception as e:
            print(cfgs, 'seed', seed, 'FAILED', out[-300:], flush=True)
