#!/bin/bash
# round 6: SAC — fused collector (scg_sac_sample / scg_sac_push), finish folded into the critics' reduction, fused critic-phase kernel; step-launch tests after the split removal
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s126; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_step_launch.py tests/test_gpu_rl.py tests/test_gpu_multirank.py tests/test_gpu_dropin.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof6; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_sac -o p -- python tools/learner_profile.py sac --iters 200 > $P/kt_sac.log 2>&1 < /dev/null
timeout 300 python tools/learner_profile.py sac --iters 200 > $P/plain_sac.log 2>&1 < /dev/null
SCG_SAC_UNFUSED_CRITIC=1 timeout 300 python tools/learner_profile.py sac --iters 200 > $P/plain_sac_unfused_critic.log 2>&1 < /dev/null
grep LEARNER_PROFILE $P/plain_sac.log $P/plain_sac_unfused_critic.log | cut -c1-400
python tools/learner_profile_post.py $P | cut -c1-1500
find $P -name '*kernel_trace.csv' -delete; find $P -name '*agent_info.csv' -delete; find $P -name '*.db' -delete
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof6/r06_kernel_stats_sac_iteration.csv')))
for r in rows[:12]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
