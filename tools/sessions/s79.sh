#!/bin/bash
# round 3, session 43 (last GPU seconds of the round): tools/candidates/r04_q2_recurrence_integrator.patch as a tagged variant
# (libscg_spec_<hash>_recur.so, built from a scratch copy of the sources: the product tree is untouched) — headline launch period
# against the shipped library on the same box, then float32 one-step errors of the variant against the oracle
cd "$GRAFT_REPO_ROOT" || exit 1
B="--steps 4000 --warmup 500 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
for tag in recur "" recur ""; do
  SCG_SPEC_TAG=$tag python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tag=[$tag]', round(d['roofline']['avg_launch_us'],4), 'us', round(d['roofline']['frac'],4), d['config']['kernel_build'], d['config']['finite_outputs'])"
done
SCG_SPEC_TAG=recur python - <<'PY'
import numpy as np, torch
from oracle.envs import make_oracle_env, make_rng
from oracle.vec import OracleVecEnv
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.vec_env import HipVecEnv
from tests.test_gpu_env_parity import _raw_state
env_id, cfg = load_task('quadrotor_2D_track')
n, seed = 512, 3
gpu = HipVecEnv(env_id, n, seed=seed, dtype=torch.float32, return_numpy=False, specialize=True, **cfg)
o = make_oracle_env(env_id, n, make_rng('philox', n, seed), **cfg); ov = OracleVecEnv(o)
ov.reset(); gpu.reset_tensors()
rng = np.random.default_rng(0); worst = np.zeros(6); bad = 0
for t in range(60):
    gpu.set_raw_state(_raw_state(o)); gpu.set_counters(o.ctrl_step_counter, o.episode)
    act = rng.uniform(-1, 1, (n, 2))
    obs_o, rew_o, done_o, info = ov.step(act)
    out = gpu.step_tensors(torch.as_tensor(act, dtype=torch.float32, device=gpu.device))
    same = out.done.cpu().numpy().astype(bool) == done_o; bad += int((~same).sum())
    keep = same & ~done_o
    worst = np.maximum(worst, np.abs(out.state.cpu().numpy().T[keep] - o.state[keep]).max(axis=0))
print('recur f32 one-step max |state error| per dim', worst, 'done mismatches', bad, 'specialised', gpu.specialized)
PY
