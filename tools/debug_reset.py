import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle.envs import make_oracle_env, make_rng
from oracle.vec import OracleVecEnv
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.vec_env import HipVecEnv
env_id, cfg = load_task('quadrotor_3D_track_disturbed')
n, seed = 65536, 5
ovec = OracleVecEnv(make_oracle_env(env_id, n, make_rng('philox', n, seed), **cfg))
gpu = HipVecEnv(env_id, n, seed=seed, dtype=torch.float64, return_numpy=False, **cfg)
a = gpu.reset_tensors().cpu().numpy(); b = ovec.reset()[0]
bad = np.argwhere(np.abs(a - b) > 1e-10 + 1e-9 * np.abs(b))
print(bad[:40]); 
for e, c in bad[:12]:
    print(e, c, a[e, c], b[e, c], a[e, :12], b[e, :12])
print('raw', np.abs(gpu.get_raw_state() - 0).shape)
for c in (14, 16):
    ua, ca = np.unique(a[:, c], return_counts=True); ub, cb = np.unique(b[:, c], return_counts=True)
    print('col', c, 'gpu', list(zip(ua, ca))[:6], 'oracle', list(zip(ub, cb))[:6])
XG = np.asarray(gpu.spec.X_GOAL); XO = np.asarray(ovec.env.X_GOAL)
print('XG shapes', XG.shape, XO.shape, 'max diff', np.abs(XG - XO).max())
print('gpu row', XG[1, :6], 'oracle row', XO[1, :6])
for v in (a[50636, 14], b[50636, 14]):
    print(v, np.argwhere(np.abs(XG - v) < 1e-12)[:4], np.argwhere(np.abs(XO - v) < 1e-12)[:4])
a2 = gpu.reset_tensors().cpu().numpy()
print('second reset bad', np.argwhere(np.abs(a2[:, 12:] - b[:, 12:]) > 1e-9)[:10])
