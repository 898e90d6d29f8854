#!/bin/bash
# rocprofv3 passes behind profiles/r04_*: run ON THE GPU BOX (gpurun -- 'bash tools/profile_round4.sh'), writes gpurun_out/prof/
# (condensed HERE by `python tools/profile_post.py r04`).  Counter passes are separate from the kernel trace and from each other.
#   step kernels (one launch per control step): the four shipped tasks at 65 536 envs f32 + the headline in float64:
#       --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_INSTS_VALU SQ_WAVES
#   K-steps-per-launch kernels (tools/seq_profile.py): step_sequence_kernel K = 8 all outputs / K = 32 collector outputs, and
#       rollout_policy_kernel (PPO's fused collector): the same four passes
#   learner iterations (PPO 16 384 / 65 536 envs, SAC): kernel trace only   (SCG_PROFILE_LEARNERS_ONLY=1: only these; SCG_PROFILE_ENV_ONLY=1: skip them)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-secondary --ppo-seeds 0 --sac-seeds 0"
if [ -z "$SCG_PROFILE_LEARNERS_ONLY" ]; then
for spec in quadrotor_2D_track:65536:f32 cartpole_stab:65536:f32 quadrotor_3D_track:65536:f32 quadrotor_3D_track_disturbed:65536:f32 quadrotor_2D_track:65536:f64; do
  IFS=: read T N DT <<< "$spec"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${T}_${DT}_$N -o p -- \
      python bench.py --task $T --envs $N --dtype $DT --steps 2000 --warmup 200 $B > $OUT/kt_${T}_${DT}_$N.log 2>&1 < /dev/null
  for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES"; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C// /+}_${T}_${DT}_$N" -o p -- \
        python bench.py --task $T --envs $N --dtype $DT --steps 100 --warmup 30 --no-graph $B > "$OUT/pmc_${C// /+}_${T}_${DT}_$N.log" 2>&1 < /dev/null
  done
done
for M in sequence_all sequence_collector rollout_policy; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${M}_f32_65536 -o p -- \
      python tools/seq_profile.py --mode $M --reps 60 > $OUT/kt_$M.log 2>&1 < /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_${M}_f32_65536 -o p -- \
        python tools/seq_profile.py --mode $M --reps 30 > $OUT/pmc_${C}_$M.log 2>&1 < /dev/null
  done
done
fi
if [ -z "$SCG_PROFILE_ENV_ONLY" ]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_iteration -o p -- \
    python tools/ppo_profile.py --fused-rollout --iters 30 --epochs 2 --minibatch 16256 > $OUT/ppo_iteration.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_iteration_65536 -o p -- \
    python tools/ppo_profile.py --fused-rollout --envs 65536 --iters 20 --epochs 2 --minibatch 16256 --mb-per-epoch 32 > $OUT/ppo_iteration_65536.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sac_iteration -o p -- \
    python tools/sac_time_to_reward.py --budget 12 --eval-every 100000 > $OUT/sac_iteration.log 2>&1 < /dev/null
fi
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*.db' -delete; find $OUT -name '*.log' -size +200k -delete
du -sh $OUT; ls $OUT | head -80
