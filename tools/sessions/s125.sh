#!/bin/bash
# round 6, first GPU session: the changed paths' tests, the learner sync-cost microbenchmark, split / wsback step A/B, learner profile, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s125; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_split_step.py tests/test_gpu_rl.py tests/test_gpu_learn.py tests/test_gpu_sac_fused.py tests/test_gpu_multirank.py -x -q -m gpu ) > $O/pytest_changed.txt 2>&1; tail -8 $O/pytest_changed.txt
timeout 120 tools/learner_sync_cost 48 50 > $O/learner_sync_cost.txt 2>&1; cat $O/learner_sync_cost.txt
# split launch (paired workgroup + barrier) vs one-wave launch, and wsback on / off
for N in 16384 32768; do
  for S in 0 32768; do
    SCG_SPLIT_MAX_ENVS=$S timeout 120 python bench.py --envs $N --steps 4000 --warmup 500 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split_max=$S N=$N', round(d['roofline']['avg_launch_us'],3),'us', d['roofline']['kernel'][:60])" >> $O/split_ab.txt
  done
done
for N in 131072 262144 524288; do
  for W in 1 999999999; do
    SCG_WSBACK_MIN_ENVS=$W timeout 120 python bench.py --envs $N --steps 2000 --warmup 300 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wsback_min=$W N=$N', round(d['roofline']['avg_launch_us'],3),'us frac', round(d['roofline']['frac'],4))" >> $O/split_ab.txt
  done
done
cat $O/split_ab.txt
bash tools/profile_round6.sh > $O/profile6.log 2>&1; tail -5 $O/profile6.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/s125/bench_default.json').read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step')}), json.dumps(d['roofline'].get('learners')))
print(json.dumps(d.get('ppo', {}).get('iteration_ms')), d.get('ppo', {}).get('wall_clock_to_two_consecutive_s'), d.get('ppo', {}).get('error'), d.get('ppo', {}).get('trace'))
print(json.dumps(d.get('sac', {}).get('roofline')), d.get('sac', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('error'), d.get('sac', {}).get('trace'))
PY
