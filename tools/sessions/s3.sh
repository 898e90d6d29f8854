#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s3; mkdir -p $O
python tools/ppo_profile.py 2>/dev/null | tail -1
python tools/ppo_profile.py --no-fused 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python tools/ppo_profile.py --iters 5 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
for seed in 1 2 3 4 5 6; do
  timeout 100 python examples/train_ppo.py --max-seconds 30 --seed $seed --quiet 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed $seed fused', d['iterations'], round(d['wall_clock_to_target_s'] or -1,2), round(d['best_eval_return'],1))"
done
