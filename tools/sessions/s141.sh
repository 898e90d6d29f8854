#!/bin/bash
# round 6 (second session): weight-image fill pipelined (W1 / W3 / biases / first W2 chunk requested before the first LDS store, later chunks one ahead),
# first tile rows requested inside the fill: learner + rollout + SAC tests (every user of mlp_fill_lds), iteration A/B, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s141; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_multirank.py tests/test_learner_golden.py tests/test_gpu_rollout_policy.py tests/test_gpu_sac_fused.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
run() { # label, env assignments...
  L=$1; shift
  env "$@" timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update']['value_loss'])"
}
for rep in 1 2; do
  run "one-tile form, direct last round " X=1
  run "accumulating form                " SCG_LEARN_MULTI_TILE=1
  run "previous library (round 5)       " SCG_LEARN_TAG=base
done 2>&1 | tee $O/ppo_ab.txt
timeout 300 python tools/learn_cost.py 2>&1 | tail -8 | tee $O/learn_cost.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof6j; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_ppo -o p -- python tools/learner_profile.py ppo --iters 40 > $P/kt_ppo.log 2>&1 < /dev/null
timeout 300 python tools/learner_profile.py ppo --iters 40 > $P/plain_ppo.log 2>&1 < /dev/null
python tools/learner_profile_post.py $P | cut -c1-600
find $P -name '*kernel_trace.csv' -delete; find $P -name '*agent_info.csv' -delete; find $P -name '*.db' -delete
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof6j/r06_kernel_stats_ppo_iteration.csv')))
for r in rows[:6]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
cd "$GRAFT_REPO_ROOT"
SCG_LEARN_TAG=timing timeout 300 python tools/learn_cost.py --timeline --mb 16256 2>&1 | tail -18 | tee gpurun_out/s141/timeline_one_tile.txt
