#!/bin/bash
# round 6 (second session): operand exchange with the operands read one block ahead: tests, A/B against the running-sum form (two commits back), timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s158; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_multirank.py tests/test_learner_golden.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
run() { L=$1; shift
  env "$@" timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update']['value_loss'])"
}
for rep in 1 2 3; do
  run "dW2 by operand exchange            " X=1
  run "running sums (two commits back)    " SCG_LEARN_TAG=prev
done 2>&1 | tee $O/ppo_ab.txt
SCG_LEARN_TAG=timing timeout 300 python tools/learn_cost.py --timeline --mb 16256 2>&1 | tail -18 | tee $O/timeline_one_tile.txt
