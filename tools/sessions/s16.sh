#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s16; mkdir -p $O
timeout 400 python tools/sac_time_to_reward.py --budget 150 --eval-every 100 > $O/sac.json 2> $O/sac.err; tail -c 3000 $O/sac.json; tail -5 $O/sac.err
