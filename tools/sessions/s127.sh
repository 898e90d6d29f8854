#!/bin/bash
# round 6: after the SAC collector / finish fold / learner micro-changes — the touched suites, learner profiles, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s127; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_multirank.py tests/test_gpu_step_launch.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
bash tools/profile_round6.sh > $O/profile6.log 2>&1; tail -4 $O/profile6.log | cut -c1-900
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/s127/bench_default.json').read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step')}), json.dumps(d['roofline'].get('learners')))
print(d.get('ppo', {}).get('wall_clock_to_two_consecutive_s'), d.get('ppo', {}).get('iterations'), d.get('ppo', {}).get('error'), d.get('ppo', {}).get('trace'))
print(d.get('ppo', {}).get('full_epochs', {}).get('wall_clock_to_two_consecutive_s'), d.get('ppo', {}).get('envs_16384', {}).get('wall_clock_to_two_consecutive_s'))
print(d.get('sac', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('param_randomised', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('error'), d.get('sac', {}).get('trace'))
PY
