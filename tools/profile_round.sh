#!/bin/bash
# rocprofv3 passes behind profiles/r02_*: run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh'), writes gpurun_out/prof/.
# Counter passes are separate from the kernel trace (and from each other: TCC slot limit), as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof; mkdir -p $OUT
B="--no-cpu-baseline --no-secondary --ppo-seeds 0"
for spec in quadrotor_2D_track:65536 quadrotor_2D_track:1048576 quadrotor_2D_track:4194304 cartpole_stab:65536 quadrotor_3D_track:65536 quadrotor_3D_track_disturbed:65536; do
  T=${spec%%:*}; N=${spec##*:}
  S=2000; [ $N -gt 1000000 ] && S=300
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${T}_$N -o p -- \
      python bench.py --task $T --envs $N --steps $S --warmup 200 $B > $OUT/kt_${T}_$N.log 2>&1 < /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_${T}_$N -o p -- \
        python bench.py --task $T --envs $N --steps 100 --warmup 30 --no-graph $B > $OUT/pmc_${C}_${T}_$N.log 2>&1 < /dev/null
  done
done
# what bounds the 4 M-env (HBM-streaming) regime: write-request mix / stalls, read mix + L2 hit rate, wave occupancy and stall split
N=4194304; T=quadrotor_2D_track
i=0
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/diag${i}_$N -o p -- \
      python bench.py --task $T --envs $N --steps 60 --warmup 20 --no-graph $B > $OUT/diag${i}_$N.log 2>&1 < /dev/null
done
# the PPO iteration (fused rollout, learner kernels, evaluation) and the K-steps-per-launch sequence kernel
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_iteration -o p -- \
    python tools/ppo_profile.py --fused-rollout --iters 30 --minibatch 65024 > $OUT/ppo_iteration.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sequence -o p -- \
    python -c "import torch, bench; torch.cuda.set_device(0); print(bench.sequence_leg(torch, 65536))" > $OUT/sequence.log 2>&1 < /dev/null
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*.db' -delete; find $OUT -name '*.log' -size +200k -delete
du -sh $OUT; ls $OUT | head -80
