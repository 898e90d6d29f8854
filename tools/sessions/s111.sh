#!/bin/bash
# round 5, GPU session 6: same-box A/B of round 4's kernel sources (tag r04src, built from commit 31ab21c's csrc) against round 5's, in-graph and under the tracer.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s111; mkdir -p $O
B="--steps 4000 --warmup 500 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
one() {  # task envs label [env assignments...]
  local task=$1 envs=$2 label=$3; shift 3
  env "$@" python bench.py --task $task --envs $envs $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-30s %9d %-12s %.4f us  frac %.4f  finite=%s' % ('$task', $envs, '$label', r['avg_launch_us'], r['frac'] or 0, d['config']['finite_outputs']))"
}
for rep in 1 2; do
  one quadrotor_2D_track 65536 r05 SCG_X=1; one quadrotor_2D_track 65536 r04src SCG_SPEC_TAG=r04src
  one quadrotor_3D_track 65536 r05 SCG_X=1; one quadrotor_3D_track 65536 r04src SCG_SPEC_TAG=r04src; one quadrotor_3D_track 65536 q3mi SCG_SPEC_TAG=q3mi
done 2>&1 | tee $O/ab_r04_vs_r05.txt
for tag in "" r04src; do
  SCG_SPEC_TAG=$tag timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_q2_${tag:-r05} -o p -- python bench.py --task quadrotor_2D_track --envs 65536 --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary --ppo-seeds 0 --sac-seeds 0 > /dev/null 2>&1
  grep step_kernel $O/kt_q2_${tag:-r05}/*kernel_stats.csv | cut -d, -f2-8 | sed "s/^/q2 rocprof ${tag:-r05}: /"
done 2>&1 | tee $O/rocprof_r04_vs_r05.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*.db' -delete
