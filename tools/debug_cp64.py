import json, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle.envs import make_oracle_env, make_rng
from oracle.vec import OracleVecEnv
from safe_control_gym_amd.vec_env import HipVecEnv
from tests.test_gpu_parity_scale import _policy, GOLDEN, _np
case, activation = 'cartpole_stab', 'leaky_relu'
g = np.load(os.path.join(GOLDEN, f'rollout_{case}.npz'))
meta = json.loads(str(g['meta_json'])); cfg = dict(meta['config']); cfg.pop('seed', None); cfg['randomized_init'] = True
pol = _policy(dict(np.load(os.path.join(GOLDEN, 'policies.npz'))), case, activation)
n, seed = 64, 31
for spec in (False, True):
    oracle = make_oracle_env(meta['task'], n, make_rng('philox', n, seed), **cfg); ovec = OracleVecEnv(oracle)
    gpu = HipVecEnv(meta['task'], n, seed=seed, dtype=torch.float32, return_numpy=False, specialize=spec, **cfg)
    obs_o, _ = ovec.reset(); obs_g = _np(gpu.reset_tensors())
    alive = np.ones(n, bool); err = np.zeros((1000, n, 4)); mag = np.zeros((1000, 4)); dones = 0
    for t in range(1000):
        obs_o, _, done_o, _ = ovec.step(pol(obs_o))
        out = gpu.step_tensors(torch.as_tensor(pol(obs_g), dtype=torch.float32, device=gpu.device))
        obs_g = _np(out.obs); done_g = out.done.cpu().numpy().astype(bool)
        alive &= done_g == done_o; cmp = alive & ~done_o; dones += done_o.sum()
        d = np.abs(oracle.state - _np(out.state).T); d[~cmp] = 0
        err[t] = d; mag[t] = np.abs(oracle.state[cmp]).max(axis=0) if cmp.any() else 0
    den = mag.max(axis=0)
    print('spec', spec, 'alive', alive.mean(), 'dones', dones, 'den', den, 'rel', err.max(axis=(0, 1)) / den)
    e = err[:, :, 3].max(axis=0); worst = np.argsort(e)[-5:]
    for w in worst:
        tt = err[:, w, 3].argmax()
        print(' env', w, 'max err', err[:, w].max(axis=0), 'at t', tt, 'state', oracle.state[w])
    print(' err by time (max over envs, dim3):', [float('%.2e' % err[a:a + 100, :, 3].max()) for a in range(0, 1000, 100)])
    print(' err by time median env dim3:', [float('%.2e' % np.median(err[a:a + 100, :, 3].max(axis=0))) for a in range(0, 1000, 100)])
