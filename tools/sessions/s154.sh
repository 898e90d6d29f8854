#!/bin/bash
# round 6 (second session): the evaluation's two workgroups on XCDs 6 and 7 (scg_policy_rollout.first_workgroup = 6: where the 254-workgroup launches of
# the training stream leave their spare CUs), started behind the collector's launch (iteration as two graphs + PPO.after_rollout): tests, the PPO loop
# alone / with the old placement / with the new one, then the driver-style bench line (PPO leg)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s154; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_rl.py tests/test_gpu_rollout_policy.py tests/test_gpu_multirank.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { L=$1; shift
  timeout 300 python tools/learner_profile.py ppo --iters 60 "$@" 2>&1 | grep -E "LEARNER_PROFILE" | python -c "
import sys, json
t = sys.stdin.read()
d = json.loads(t.split('LEARNER_PROFILE ')[1].splitlines()[0]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4))"
}
for rep in 1 2; do
  run "no evaluation                                            "
  run "evaluation on XCDs 0, 1, behind train_step (before)      " --eval-chunk 0 --eval-behind --eval-first-workgroup 0
  run "evaluation on XCDs 6, 7, behind train_step               " --eval-chunk 0 --eval-behind --eval-first-workgroup 6
  run "evaluation on XCDs 0, 1, behind the collector launch     " --eval-chunk 0 --eval-first-workgroup 0
  run "evaluation on XCDs 6, 7, behind the collector launch     " --eval-chunk 0 --eval-first-workgroup 6
done 2>&1 | tee $O/eval_interference.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --sac-seeds 0 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 200 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s154/bench_driver.json').read().strip().splitlines()[-1])
print(json.dumps({k: v for k, v in d['ppo']['iteration_ms'].items() if k != 'what'}))
print(d['ppo'].get('wall_clock_to_two_consecutive_s'), d['ppo'].get('iterations'), d['ppo'].get('error'))
PY
