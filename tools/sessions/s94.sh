#!/bin/bash
# round 4, GPU session 12 (≈ 6 GPU-minutes): CartPole substep — Bullet's velocity clamp decided once per control step for the whole wave
# (dropped from the 50 substeps when it provably cannot trigger) + (m l sin, m l cos) as one packed product.  Parity files that step
# CartPole, then same-box A/B against the previous kernel (tag cpold), incl. scg_rollout_random (BASELINE config #2's workload).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s94; mkdir -p $O
( time timeout 600 python -m pytest -m gpu -q tests/test_gpu_env_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_config_fuzz.py tests/test_gpu_sequence.py tests/test_gpu_facade.py ) > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
timeout 300 python tools/ab_variant.py run cpold --tasks cartpole_stab --rounds 2 2>&1 | tee $O/ab_cpold.log | grep 'tag=\|one-step' | cut -c1-250
for tag in cpold "" cpold ""; do
SCG_SPEC_TAG=$tag python - <<'PY' 2>&1 | tail -1
import os, time, torch
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.vec_env import HipVecEnv
torch.cuda.set_device(0)
env_id, cfg = load_task('cartpole_stab')
env = HipVecEnv(env_id, 65536, seed=1, return_numpy=False, **cfg)
env.reset_tensors()
env.rollout_random(1000); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    env.rollout_random(1000)
torch.cuda.synchronize()
print('rollout_random tag=[%s]' % os.environ.get('SCG_SPEC_TAG', ''), '%.4g env-steps/s' % (3 * 1000 * 65536 / (time.perf_counter() - t0)))
PY
done
