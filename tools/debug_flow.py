import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_gym_amd.ppo import PPO, PPOConfig, evaluate
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.vec_env import HipVecEnv
torch.cuda.set_device(0)
fused, do_eval = sys.argv[1] == 'fused', sys.argv[2] == 'eval'
env_id, cfg = load_task('quadrotor_2D_track')
N, T = 16384, 32
env = HipVecEnv(env_id, N, seed=3, return_numpy=False, **cfg)
eval_env = HipVecEnv(env_id, 256, seed=333, return_numpy=False, **dict(cfg, randomized_init=False))
pc = dict(hidden_dim=128, activation='tanh', use_gae=True, target_kl=0.03, opt_epochs=4, mini_batch_size=65536, actor_lr=2e-3,
          critic_lr=2e-3, rollout_batch_size=N, rollout_steps=T)
ppo = PPO(env, PPOConfig(**pc, extra={'fused_update': fused}), seed=3)
for it in range(4):
    res = ppo.train_step()
    rng = {k: (round(float(v.min()), 3), round(float(v.max()), 3)) for k, v in (('logp', ppo.logp), ('act', ppo.act), ('obs', ppo.obs), ('v', ppo.v))}
    print(sys.argv[1:], it, {k: res[k] for k in ('policy_loss', 'value_loss', 'approx_kl', 'actor_steps')}, rng, flush=True)
    if do_eval:
        print('   eval', evaluate(ppo.agent.ac, eval_env)['ep_return'])
