#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s30; mkdir -p $O
run() { tag=$1; shift; python tools/ppo_seeds.py --seeds 6 "$@" > $O/$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/$tag.json')); t=d['wall_clock_to_target_s']; print('$tag', 'median', round(d['median_s'],3) if d['median_s'] else None, 'reached', d['reached'], [round(x,2) if x else None for x in t], d['iterations'])"; }
run N16k_mb16k --budget 6 --envs 16384 --minibatch 16256
run N16k_mb32k_again --budget 6 --envs 16384 --minibatch 32512
run N16k_mb32k_ep3 --budget 6 --envs 16384 --minibatch 32512 --epochs 3
run N65k_mb32k --budget 12 --envs 65536 --minibatch 32512
run N65k_mb130k --budget 12 --envs 65536 --minibatch 130048
