#!/bin/bash
# round 6 (second session): critic pass on 31 workgroups per XCD (one CU per XCD left free) — the PPO loop with the asynchronous evaluation beside it,
# this library vs the previous commit's (254 workgroups), and without evaluation
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s155; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_learn.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { L=$1; shift
  env "$@" timeout 300 python tools/learner_profile.py ppo --iters 60 $EXTRA 2>&1 | grep -E "LEARNER_PROFILE" | python -c "
import sys, json
t = sys.stdin.read()
d = json.loads(t.split('LEARNER_PROFILE ')[1].splitlines()[0]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4))"
}
for rep in 1 2; do
  EXTRA="" run "no evaluation, 248-workgroup critic pass       " X=1
  EXTRA="" run "no evaluation, 254 (previous commit)            " SCG_LEARN_TAG=prev
  EXTRA="--eval-chunk 0" run "evaluation beside it, 248                       " X=1
  EXTRA="--eval-chunk 0" run "evaluation beside it, 254 (previous commit)     " SCG_LEARN_TAG=prev
done 2>&1 | tee $O/eval_interference.txt
