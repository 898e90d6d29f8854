#!/bin/bash
# round 5, second round of PPO leg probes (epochs x minibatches per epoch) at 65 536 envs, 8 seeds each (bench.py's protocol and target; tools/ppo_seeds.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s119; mkdir -p $O
probe() {  # label args...
  local label=$1; shift
  timeout 200 python tools/ppo_seeds.py --envs 65536 --seeds 8 --budget 6 "$@" 2>/dev/null | tail -1 > $O/$label.json
  python -c "
import json; d=json.load(open('$O/$label.json')); t=d['wall_clock_to_two_consecutive_s']; ok=sorted(x for x in t if x is not None)
print('%-28s reached %d/8  median %s  mean-of-reached %.3f  iters %s' % ('$label', len(ok), d['median_s'], sum(ok)/max(1,len(ok)), d['iterations']))"
}
probe base_2x32 --mb-per-epoch 32
probe mb16x3 --mb-per-epoch 16 --epochs 3
probe mb16x2 --mb-per-epoch 16 --epochs 2
probe mb12x3 --mb-per-epoch 12 --epochs 3
probe mb12x4 --mb-per-epoch 12 --epochs 4
probe mb8x4 --mb-per-epoch 8 --epochs 4
probe mb16x4 --mb-per-epoch 16 --epochs 4
probe mb20x3 --mb-per-epoch 20 --epochs 3
