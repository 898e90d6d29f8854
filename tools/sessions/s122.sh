#!/bin/bash
# round 5: mixed store policy — workspace arrays (state, counters: re-read by the next launch) written back, outputs written through
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s122; mkdir -p $O
B="--steps 4000 --warmup 500 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
one() {
  local task=$1 envs=$2 label=$3; shift 3
  env "$@" python bench.py --task $task --envs $envs $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s %9d %-10s %.4f us  frac %.4f  finite=%s' % ('$task', $envs, '$label', r['avg_launch_us'], r['frac'] or 0, d['config']['finite_outputs']))"
}
for rep in 1 2; do for tag in "" ws0 ws1; do one quadrotor_2D_track 65536 "tag=$tag" SCG_SPEC_TAG=$tag; done; done 2>&1 | tee $O/ab_ws.txt
for tag in "" ws0; do one quadrotor_2D_track 262144 "tag=$tag" SCG_SPEC_TAG=$tag; done 2>&1 | tee -a $O/ab_ws.txt
