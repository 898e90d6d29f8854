#!/usr/bin/env python3
"""One control step of a fuzzed config (tests/config_fuzz.py) on the float64 generic kernels vs the oracle, per-env deltas.
usage: fuzz_diag.py <system> <seed> [steps]      (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleVecEnv
    from safe_control_gym_amd.vec_env import HipVecEnv
    from tests.config_fuzz import fuzz_config
    system, seed = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    env_id, cfg = fuzz_config(system, seed)
    for k, v in (a.split('=') for a in sys.argv[4:]):
        cfg[k] = eval(v)
    n = 70
    o = make_oracle_env(env_id, n, make_rng('philox', n, 100 + seed), **cfg)
    ov = OracleVecEnv(o)
    g = HipVecEnv(env_id, n, seed=100 + seed, dtype=torch.float64, return_numpy=False, specialize=False, **cfg)
    ov.reset(); g.reset_tensors()
    rng = np.random.default_rng(seed)
    for t in range(steps):
        act = rng.uniform(-1.2, 1.2, (n, o.action_dim))
        if not o.NORMALIZED_RL_ACTION_SPACE:
            lo, hi = o.physical_action_bounds
            act = lo + (act + 1.2) / 2.4 * (hi - lo) * 1.1 - 0.05 * (hi - lo)
        adv = None
        if o.adversary_disturbance is not None:
            a = rng.uniform(-1.3, 1.3, (n, o.adversary_dim))
            o.set_adversary_control(a); g.set_adversary_control(a); adv = g._adv
        s0 = o.state.copy()
        obs_o, rew_o, done_o, info = ov.step(act)
        out = g.step_tensors(torch.as_tensor(act, dtype=torch.float64, device=g.device), adv)
        g._adv = None
        st = out.state.cpu().numpy().T
        d = st - o.state
        k = int(np.argmax(np.abs(d).max(axis=1)))
        print(f't={t} max|dstate|={np.abs(d).max():.3e} env {k}: before {s0[k]} oracle {o.state[k]} hip {st[k]} delta {d[k]} done {done_o[k]}')
        print('   per-dim max', np.abs(d).max(axis=0), 'n_envs off', int((np.abs(d).max(axis=1) > 1e-12).sum()))


if __name__ == '__main__':
    main()
