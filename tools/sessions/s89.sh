#!/bin/bash
# round 4, ninth GPU session (≈ 5 GPU-minutes): evaluation with the fused deterministic SAC actor + packed accumulators — the RL test files,
# then the SAC legs of the bench (PPO legs off).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s89; mkdir -p $O
( time timeout 600 python -m pytest -m gpu -q tests/test_gpu_rl.py tests/test_gpu_sac_fused.py tests/test_gpu_dropin.py tests/test_gpu_adversarial.py tests/test_gpu_facade.py ) > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
( time python bench.py --gpus 1 --steps 20 --warmup 5 --ppo-seeds 0 --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s89/bench.json').read().strip().splitlines()[-1])
s = d['sac']
print('sac', s.get('error'), s.get('wall_clock_to_two_consecutive_s'), s.get('gradient_steps'), s.get('env_steps_per_s_incl_learning'))
p = s.get('param_randomised', {})
print('param_randomised', p.get('error'), p.get('wall_clock_to_two_consecutive_s'), p.get('env_steps_per_s_incl_learning'), p.get('target_return'))
PY
