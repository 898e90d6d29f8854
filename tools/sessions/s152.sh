#!/bin/bash
# round 6 (second session): the iteration as two graphs (collector launch | everything behind it) with the evaluation enqueued between them: RL tests, the
# PPO loop alone / evaluation behind train_step / evaluation between the replays, then the driver-style bench line (PPO leg only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s152; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_rl.py tests/test_gpu_rollout_policy.py tests/test_gpu_multirank.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { L=$1; shift
  timeout 300 python tools/learner_profile.py ppo --iters 60 "$@" 2>&1 | grep -E "LEARNER_PROFILE" | python -c "
import sys, json
t = sys.stdin.read()
d = json.loads(t.split('LEARNER_PROFILE ')[1].splitlines()[0]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4))"
}
for rep in 1 2; do
  run "no evaluation                                  "
  run "evaluation enqueued behind train_step          " --eval-chunk 0 --eval-behind
  run "evaluation enqueued between the two replays    " --eval-chunk 0
done 2>&1 | tee $O/eval_interference.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --sac-seeds 0 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 200 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s152/bench_driver.json').read().strip().splitlines()[-1])
print(json.dumps({k: v for k, v in d['ppo']['iteration_ms'].items() if k != 'what'}))
print(d['ppo'].get('wall_clock_to_two_consecutive_s'), d['ppo'].get('iterations'), d['ppo'].get('error'))
PY
