#!/bin/bash
# round 5: one-wave vs 256-thread workgroups at 1 M ... 16 M envs on ONE box, short and long runs (same library: SCG_WIDE_MIN_ENVS decides)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s116; mkdir -p $O
one() {  # task envs steps label [env assignments...]
  local task=$1 envs=$2 steps=$3 label=$4; shift 4
  env "$@" python bench.py --task $task --envs $envs --steps $steps --warmup $(( steps / 10 )) --graph-len $(( steps < 1000 ? steps : 1000 )) --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s %9d steps %5d %-6s %.2f us  frac %.4f' % ('$task', $envs, $steps, '$label', r['avg_launch_us'], r['frac'] or 0))"
}
for rep in 1 2; do
for n in 1048576 2097152 4194304 8388608 16777216; do
  steps=$(( 2000000000 / n )); [ $steps -gt 2000 ] && steps=2000
  one quadrotor_2D_track $n $steps one-wave SCG_WIDE_MIN_ENVS=2000000000; one quadrotor_2D_track $n $steps wide SCG_WIDE_MIN_ENVS=1
done; done 2>&1 | tee $O/wide_vs_one_wave.txt
one quadrotor_2D_track 16777216 3000 one-wave SCG_WIDE_MIN_ENVS=2000000000 | tee -a $O/wide_vs_one_wave.txt
one quadrotor_2D_track 16777216 3000 wide SCG_WIDE_MIN_ENVS=1 | tee -a $O/wide_vs_one_wave.txt
for n in 4194304 16777216; do one cartpole_stab $n 200 one-wave SCG_WIDE_MIN_ENVS=2000000000; one cartpole_stab $n 200 wide SCG_WIDE_MIN_ENVS=1; done 2>&1 | tee -a $O/wide_vs_one_wave.txt
