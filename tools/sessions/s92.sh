#!/bin/bash
# round 4, GPU session 10 (≈ 5 GPU-minutes): scg_ppo_step — reduction + gated Adam in one fence-free launch (single rank) — tests, then the
# steady-state iteration time with and without it, then the PPO legs of the bench.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s92; mkdir -p $O
( time timeout 600 python -m pytest -m gpu -q tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_adversarial.py tests/test_gpu_multirank.py ) > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
python tools/ppo_iter_times.py 2>&1 | grep seed | cut -c1-250
python tools/ppo_iter_times.py --no-fused-step 2>&1 | grep seed | cut -c1-250
python tools/ppo_iter_times.py 2>&1 | grep seed | cut -c1-250
( time python bench.py --gpus 1 --steps 20 --warmup 5 --sac-seeds 0 --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s92/bench.json').read().strip().splitlines()[-1])
r = d['ppo']
print('ppo', r.get('error'), r.get('wall_clock_to_two_consecutive_s'), r.get('iterations'), 'median', r.get('median_s'), '16384:', r.get('envs_16384', {}).get('wall_clock_to_two_consecutive_s'),
      'full:', r.get('full_epochs', {}).get('wall_clock_to_two_consecutive_s'))
PY
