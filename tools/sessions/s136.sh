#!/bin/bash
# round 6 (second session): SAC wide kernels — row index unconditional (no round trip at kernel entry), db3 / statistics through the LDS instead of
# ds_bpermute butterflies, lane-half exchanges batched, data-gradient operand and the input gather requested last, reward / mask by batch row
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s136; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_rl.py tests/test_gpu_multirank.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python tools/sac_step_ab.py shipped base > $O/sac_ab.txt 2>&1; tail -5 $O/sac_ab.txt
timeout 300 python tools/sac_timeline.py > $O/sac_timeline.txt 2>&1; tail -13 $O/sac_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof6e; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_sac -o p -- python tools/learner_profile.py sac --iters 200 > $P/kt_sac.log 2>&1 < /dev/null
timeout 300 python tools/learner_profile.py sac --iters 200 > $P/plain_sac.log 2>&1 < /dev/null
python tools/learner_profile_post.py $P | cut -c1-400
find $P -name '*kernel_trace.csv' -delete; find $P -name '*agent_info.csv' -delete; find $P -name '*.db' -delete
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof6e/r06_kernel_stats_sac_iteration.csv')))
for r in rows[:9]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
