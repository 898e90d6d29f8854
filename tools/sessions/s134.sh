#!/bin/bash
# round 6 (second session): SAC step A/B against the previous commit's library on one box (hash of the state after 80 steps + us per step), timeline of
# the new actor_grad_kernel, kernel trace of the SAC leg; PPO reduction + Adam kernel with every load in flight: learner tests + A/B of the iteration
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s134; mkdir -p $O
timeout 600 python tools/sac_step_ab.py shipped base > $O/sac_ab.txt 2>&1; tail -6 $O/sac_ab.txt
timeout 300 python tools/sac_timeline.py > $O/sac_timeline.txt 2>&1; tail -13 $O/sac_timeline.txt
( time timeout 900 python -m pytest tests/test_gpu_learn.py -x -q -m gpu ) > $O/pytest_learn.txt 2>&1; tail -3 $O/pytest_learn.txt
for T in "" base "" base; do
  SCG_LEARN_TAG=$T timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('tag [$T]', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update'])"
done 2>&1 | tee $O/ppo_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof6d; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_sac -o p -- python tools/learner_profile.py sac --iters 200 > $P/kt_sac.log 2>&1 < /dev/null
timeout 300 python tools/learner_profile.py sac --iters 200 > $P/plain_sac.log 2>&1 < /dev/null
python tools/learner_profile_post.py $P | cut -c1-400
find $P -name '*kernel_trace.csv' -delete; find $P -name '*agent_info.csv' -delete; find $P -name '*.db' -delete
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof6d/r06_kernel_stats_sac_iteration.csv')))
for r in rows[:9]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
