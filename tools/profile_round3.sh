#!/bin/bash
# rocprofv3 passes behind profiles/r03_*: run ON THE GPU BOX (gpurun -- 'bash tools/profile_round3.sh'), writes gpurun_out/prof/
# (condensed by tools/profile_post.py r03).  Counter passes are separate from the kernel trace and from each other.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-secondary --ppo-seeds 0 --sac-seeds 0"
for spec in quadrotor_2D_track:65536 cartpole_stab:65536 quadrotor_3D_track:65536 quadrotor_3D_track_disturbed:65536; do
  T=${spec%%:*}; N=${spec##*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${T}_$N -o p -- \
      python bench.py --task $T --envs $N --steps 2000 --warmup 200 $B > $OUT/kt_${T}_$N.log 2>&1 < /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_${T}_$N -o p -- \
        python bench.py --task $T --envs $N --steps 100 --warmup 30 --no-graph $B > $OUT/pmc_${C}_${T}_$N.log 2>&1 < /dev/null
  done
done
# the PPO iteration at 16 384 envs (32-envs-per-wave rollout) and at 65 536 envs, the SAC iteration (fused update)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_iteration -o p -- \
    python tools/ppo_profile.py --fused-rollout --iters 30 --minibatch 65024 > $OUT/ppo_iteration.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_iteration_65536 -o p -- \
    python tools/ppo_profile.py --fused-rollout --envs 65536 --iters 12 --minibatch 32512 > $OUT/ppo_iteration_65536.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sac_iteration -o p -- \
    python tools/sac_time_to_reward.py --budget 12 --eval-every 100000 > $OUT/sac_iteration.log 2>&1 < /dev/null
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*.db' -delete; find $OUT -name '*.log' -size +200k -delete
du -sh $OUT; ls $OUT | head -80
