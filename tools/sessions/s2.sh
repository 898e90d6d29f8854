#!/bin/bash
# GPU session 2: pinning tests on GPU, fused learner kernels vs autograd, PPO iteration timing fused vs torch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_learn.py tests/test_gpu_symbolic.py tests/test_bullet_convergence.py -m gpu -q ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log
for mode in "" "--no-fused"; do
  timeout 120 python examples/train_ppo.py --max-seconds 40 --seed 2 --quiet $mode > $O/ppo$mode.json 2> $O/ppo$mode.err
  tail -1 $O/ppo$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', {k:d[k] for k in ('iterations','env_steps','wall_clock_s','wall_clock_to_target_s','best_eval_return')})"
  tail -3 $O/ppo$mode.err
done
