#!/bin/bash
# round 6 (second session): the asynchronous evaluation as 25-step launches (a collector workgroup waits for one chunk at most, not for the whole 2.9 ms
# evaluation): test, then the driver-style bench line (PPO leg: device / wall per iteration beside the evaluation)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s149; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_rollout_policy.py tests/test_gpu_rl.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --sac-seeds 0 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 200 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s149/bench_driver.json').read().strip().splitlines()[-1])
print(json.dumps(d['ppo']['iteration_ms']))
print(d['ppo'].get('wall_clock_to_two_consecutive_s'), d['ppo'].get('iterations'))
PY
