#!/bin/bash
# round 6 (second session): the asynchronous evaluation as ONE captured graph per evaluation (host cost of launch()): the PPO loop alone, with the eager
# evaluation, with the captured one; RL tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s151; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_rl.py tests/test_gpu_rollout_policy.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { L=$1; shift
  timeout 300 python tools/learner_profile.py ppo --iters 60 "$@" 2>&1 | grep -E "LEARNER_PROFILE|Warning|Error" | python -c "
import sys, json
t = sys.stdin.read()
if 'LEARNER_PROFILE ' not in t: print('$L', 'FAILED', t[-300:]); sys.exit()
d = json.loads(t.split('LEARNER_PROFILE ')[1].splitlines()[0]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), t[:200] if 'arn' in t.split('LEARNER_PROFILE')[0] else '')"
}
for rep in 1 2; do
  run "no evaluation                     "
  run "evaluation, eager sequence        " --eval-chunk 0 --eval-eager
  run "evaluation, one captured graph    " --eval-chunk 0
done 2>&1 | tee $O/eval_interference.txt
