#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s29; mkdir -p $O
run() { tag=$1; shift; python tools/ppo_seeds.py --envs 16384 --minibatch 65024 --seeds 6 --budget 6 "$@" > $O/$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/$tag.json')); t=d['wall_clock_to_target_s']; print('$tag', 'median', d['median_s'], 'reached', d['reached'], [round(x,2) if x else None for x in t], d['iterations'])"; }
run base
run kl05 --target-kl 0.05
run kl1 --target-kl 0.1
run lr3 --lr 3e-3
run lr3kl05 --lr 3e-3 --target-kl 0.05
run ep3 --epochs 3
run ep6 --epochs 6
