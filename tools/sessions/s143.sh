#!/bin/bash
# round 6 (second session): the collector's post-processing as library launches (scg_ppo_returns_*): tests of every PPO path, iteration A/B
# (SCG_LEARN_TAG=base still runs: the old library lacks the entry points, so that row is skipped if it fails), rocprofv3 of the iteration
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s143; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_multirank.py tests/test_gpu_rollout_policy.py tests/test_gpu_adversarial.py tests/test_gpu_dropin.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2 3; do
  timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('returns as library launches', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update']['value_loss'])"
done 2>&1 | tee $O/ppo_ab.txt
bash tools/profile_round6.sh > $O/profile6.log 2>&1; tail -4 $O/profile6.log | cut -c1-600
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof6/r06_kernel_stats_ppo_iteration.csv')))
for r in rows[:12]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
