#!/bin/bash
# round 6 (second session): what the asynchronous evaluation costs the training stream — the PPO iteration loop alone, with the evaluation as one launch
# per evaluation, and as 25-step launches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s150; mkdir -p $O
run() { L=$1; shift
  timeout 300 python tools/learner_profile.py ppo --iters 60 "$@" 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4))"
}
for rep in 1 2; do
  run "no evaluation                "
  run "evaluation, one launch       " --eval-chunk 0
  run "evaluation, 25-step launches " --eval-chunk 25
  run "evaluation, 5-step launches  " --eval-chunk 5
done 2>&1 | tee $O/eval_interference.txt
