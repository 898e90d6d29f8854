#!/bin/bash
# round 5: workspace arrays written back (tag ws0; ws0w = + 256-thread workgroups) by shard size, same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s123; mkdir -p $O
one() {  # task envs steps label [env assignments...]
  local task=$1 envs=$2 steps=$3 label=$4; shift 4
  env "$@" python bench.py --task $task --envs $envs --steps $steps --warmup $(( steps / 10 )) --graph-len $(( steps < 1000 ? steps : 1000 )) --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-20s %9d %-9s %.2f us  frac %.4f' % ('$task', $envs, '$label', r['avg_launch_us'], r['frac'] or 0))"
}
P="SCG_WIDE_MIN_ENVS=2000000000"
for n in 98304 131072 524288 1048576 4194304; do
  steps=$(( 1000000000 / n )); [ $steps -gt 3000 ] && steps=3000
  one quadrotor_2D_track $n $steps shipped $P; one quadrotor_2D_track $n $steps ws0 SCG_SPEC_TAG=ws0 $P
done 2>&1 | tee $O/ab_ws_n.txt
one quadrotor_2D_track 4194304 238 ws0w SCG_SPEC_TAG=ws0w $P | tee -a $O/ab_ws_n.txt
one quadrotor_2D_track 16777216 60 shipped-w SCG_X=1 | tee -a $O/ab_ws_n.txt
one quadrotor_2D_track 16777216 60 ws0w SCG_SPEC_TAG=ws0w $P | tee -a $O/ab_ws_n.txt
for t in cartpole_stab quadrotor_3D_track; do for n in 262144 1048576; do one $t $n 1000 shipped $P; one $t $n 1000 ws0 SCG_SPEC_TAG=ws0 $P; done; done 2>&1 | tee -a $O/ab_ws_n.txt
