#!/bin/bash
# round 5: PPO leg hyper-parameter probes at 65 536 envs, 8 seeds each (bench.py's protocol and target; tools/ppo_seeds.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s118; mkdir -p $O
probe() {  # label args...
  local label=$1; shift
  timeout 200 python tools/ppo_seeds.py --envs 65536 --seeds 8 --budget 6 "$@" 2>/dev/null | tail -1 > $O/$label.json
  python -c "
import json; d=json.load(open('$O/$label.json')); t=d['wall_clock_to_two_consecutive_s']; ok=sorted(x for x in t if x is not None)
print('%-28s reached %d/8  median %s  mean-of-reached %.3f  iters %s' % ('$label', len(ok), d['median_s'], sum(ok)/max(1,len(ok)), d['iterations']))"
}
probe base_2x32 --mb-per-epoch 32
probe mb24 --mb-per-epoch 24
probe mb16x3 --mb-per-epoch 16 --epochs 3
probe lr3e-3 --mb-per-epoch 32 --lr 3e-3
probe T16 --mb-per-epoch 32 --rollout-steps 16
probe mini8128 --mb-per-epoch 32 --minibatch 8128
probe kl05 --mb-per-epoch 32 --target-kl 0.05
probe mb24_lr3e-3 --mb-per-epoch 24 --lr 3e-3
