#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for seed in 1 3 4; do
  timeout 100 python examples/train_ppo.py --max-seconds 30 --seed $seed --quiet --no-fused 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed $seed torch', d['iterations'], round(d['wall_clock_to_target_s'] or -1,2), round(d['best_eval_return'],1))"
done
