#!/bin/bash
# round 6 (second session): critic pass requests the next tile's rows before the current tile's products: A/B of the iteration against the previous commit's library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s146; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_learn.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { L=$1; shift
  env "$@" timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update']['value_loss'])"
}
for rep in 1 2 3; do
  run "rows one tile ahead in the critic pass " X=1
  run "previous commit                        " SCG_LEARN_TAG=prev
done 2>&1 | tee $O/ppo_ab.txt
