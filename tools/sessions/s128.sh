#!/bin/bash
# round 6: (a) EXPERIMENT c_values stored as per-env rows (16-byte stores via the LDS transpose) vs the shipped SoA rows, streaming regime;
# (b) SAC reduce with the bookkeeping inputs prefetched; (c) PPO reduce unroll back at 8
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s128; mkdir -p $O
timeout 900 python tools/ab_variant.py run cvrows --tasks quadrotor_2D_track --rounds 2 --envs 65536,1048576,4194304,16777216 --no-gate > $O/cvrows_ab.txt 2>&1; cat $O/cvrows_ab.txt | tail -20
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof6b; rm -rf $P; mkdir -p $P
for M in ppo sac; do
  IT=40; [ $M = sac ] && IT=200
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_$M -o p -- python tools/learner_profile.py $M --iters $IT > $P/kt_$M.log 2>&1 < /dev/null
done
python tools/learner_profile_post.py $P | cut -c1-700
find $P -name '*kernel_trace.csv' -delete; find $P -name '*agent_info.csv' -delete; find $P -name '*.db' -delete
python - <<'PY'
import csv
for m in ('ppo', 'sac'):
    rows = list(csv.DictReader(open(f'gpurun_out/prof6b/r06_kernel_stats_{m}_iteration.csv')))
    for r in rows[:8]:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
