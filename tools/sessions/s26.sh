#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s26; mkdir -p $O
timeout 200 python tools/sac_time_to_reward.py --budget 150 --eval-every 100 --batch 16384 > $O/sac_b16k.json 2> $O/sac1.err; python -c "
import json; d=json.load(open('$O/sac_b16k.json')); r=d['runs'][0]; print('b16k', r['wall_clock_to_target_s'], r['best_eval_return'], r['vector_steps'], r['curve_s_steps_return_length'][-4:])"
timeout 200 python tools/sac_time_to_reward.py --budget 150 --eval-every 100 --batch 16384 --lr 3e-3 > $O/sac_b16k_lr3.json 2> $O/sac2.err; python -c "
import json; d=json.load(open('$O/sac_b16k_lr3.json')); r=d['runs'][0]; print('b16k lr3e-3', r['wall_clock_to_target_s'], r['best_eval_return'], r['vector_steps'], r['curve_s_steps_return_length'][-4:])"
