#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s10; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_rollout_policy.py tests/test_gpu_learn.py -m gpu -q -x ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log
python tools/ppo_profile.py --fused-rollout 2>&1 | tail -2
for seed in 1 2 3 4; do
  timeout 100 python examples/train_ppo.py --max-seconds 20 --seed $seed --quiet 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed $seed', d['iterations'], round(d['wall_clock_to_target_s'] or -1,2), round(d['best_eval_return'],1), round(d['wall_clock_s']/d['iterations']*1e3,1),'ms/it')"
done
