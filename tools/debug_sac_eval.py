import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from safe_control_gym_amd.ppo import evaluate
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.sac import MLPActorCritic
from safe_control_gym_amd.vec_env import HipVecEnv
from tools.sac_time_to_reward import Deterministic
f = np.load('tests/golden/sac_actor_quadrotor_3D_track.npz')
for task in ('quadrotor_3D_track', 'quadrotor_3D_track_disturbed'):
    env_id, cfg = load_task(task)
    for rinit in (False, True):
        env = HipVecEnv(env_id, 256, seed=4242, return_numpy=False, **dict(cfg, randomized_init=rinit))
        spec = env.spec
        low = torch.as_tensor(spec.action_space.low, dtype=torch.float32, device=env.device); high = torch.as_tensor(spec.action_space.high, dtype=torch.float32, device=env.device)
        ac = MLPActorCritic(spec.obs_dim, spec.nu, low, high, [128, 128], 'relu').to(env.device)
        ac.load_state_dict({k: torch.as_tensor(f[k]) for k in f.files if k.startswith('actor.')}, strict=False)
        ev = evaluate(Deterministic(ac), env)
        print(task, rinit, {k: ev[k] for k in ('episodes', 'ep_return', 'ep_length', 'ep_mse', 'ep_constraint_violation')}, low.tolist(), high.tolist())
        obs = env.reset_tensors()
        for t in range(3):
            a = ac.act(obs, deterministic=True); out = env.step_tensors(a); obs = out.obs
            print('  t', t, 'act', a[0].tolist(), 'obs', obs[0, :12].tolist(), 'done', bool(out.done[0]))
        env.close()
