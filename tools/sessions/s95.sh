#!/bin/bash
# round 4, GPU session 13 (≈ 3 GPU-minutes): machine-scheduler strategies for the kernels that keep LLVM's default (plain Quadrotor3D, CartPole),
# re-checked under write-through stores (round 3 chose them under write-back).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s95; mkdir -p $O
for v in "q3maxilp quadrotor_3D_track" "q3iter quadrotor_3D_track" "q3clause quadrotor_3D_track" "cpiter cartpole_stab" "cpmaxilp cartpole_stab"; do
  set -- $v
  [ -f safe_control_gym_amd/spec/*_$1.so ] || { echo "$1: not built"; continue; }
  timeout 200 python tools/ab_variant.py run $1 --tasks $2 --rounds 2 --no-gate 2>&1 | tee $O/ab_$1.log | grep tag= | cut -c1-200
done
