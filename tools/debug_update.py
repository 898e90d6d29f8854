"""Fused vs torch update on REAL rollout data of iteration 2 (same parameters, moments, data, permutations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_gym_amd.ppo import PPO, PPOConfig, PPOAgent
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.vec_env import HipVecEnv
torch.cuda.set_device(0)
env_id, cfg = load_task('quadrotor_2D_track')
N, T = 16384, 32
env = HipVecEnv(env_id, N, seed=3, return_numpy=False, **cfg)
pc = dict(hidden_dim=128, activation='tanh', use_gae=True, target_kl=0.03, opt_epochs=4, mini_batch_size=65536, actor_lr=2e-3,
          critic_lr=2e-3, rollout_batch_size=N, rollout_steps=T)
ppo = PPO(env, PPOConfig(**pc, extra={}), seed=3)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    ppo.train_step()
# iteration-2 data
ppo._rollout_graph.replay()
ret, adv, moments = ppo._rollout_out
mean = moments[0] / moments[2]; std = torch.sqrt(torch.clamp(moments[1] / moments[2] - mean * mean, min=0.0))
adv = (adv - mean) / (std + 1e-6)
M = T * N
data = {'obs': ppo.obs[:T].reshape(M, -1).clone(), 'act': ppo.act.reshape(M, -1).clone(), 'logp': ppo.logp.reshape(M).clone(),
        'adv': adv.reshape(M).clone(), 'ret': ret.reshape(M).clone(), 'v': ppo.v.reshape(M).clone()}
print('data ranges', {k: (float(v.min()), float(v.max())) for k, v in data.items()})
fl = ppo.agent._flat
state = {k: fl[k].clone() for k in ('p', 'm', 'v', 'steps')}
out = {}
for mode, extra in (('fused', {}), ('torch', {'fused_update': False})):
    ag = PPOAgent(12, 2, PPOConfig(**pc, extra=extra), 'cuda:0')
    for k in state:
        ag._flat[k].copy_(state[k])
    gen = torch.Generator(device='cuda').manual_seed(5)
    res = ag.update({k: v.clone() for k, v in data.items()}, generator=gen)
    torch.cuda.synchronize()
    out[mode] = (res, ag._flat['p'].clone(), ag._flat['m'].clone(), ag._flat['v'].clone(), ag._flat['steps'].clone())
    print(mode, res, 'logstd', ag.ac.actor.logstd.tolist())
d = (out['fused'][1] - out['torch'][1]).abs()
print('max |dp|', d.max().item(), 'at', d.argmax().item(), 'n_a', fl['n_a'], 'steps', out['fused'][4].tolist(), out['torch'][4].tolist())
print('max |dm|', (out['fused'][2] - out['torch'][2]).abs().max().item(), 'max |dv|', (out['fused'][3] - out['torch'][3]).abs().max().item())
