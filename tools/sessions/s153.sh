#!/bin/bash
# round 6 (second session): which kernels of the PPO iteration get slower beside the asynchronous evaluation — rocprofv3 kernel trace of the loop with it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s153; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python tools/learner_profile.py ppo --iters 60 --eval-chunk 0 > $O/kt.log 2>&1 < /dev/null
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s153/kt/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} min {float(r['MinNs'])/1e3:9.2f} max {float(r['MaxNs'])/1e3:9.2f} us")
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*.db' -delete
