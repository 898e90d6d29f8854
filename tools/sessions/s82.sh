#!/bin/bash
# round 4, second GPU session (≈ 10 GPU-minutes).  s81 found the write-through store policy (sc0 sc1) worth 16 % of the headline launch
# (5.24 -> 4.40 us) and 13 % of Quadrotor3D's; the product now DEFAULTS to it (SCG_ST_AUX=17, scg_env_core.h).  Here:
#   1. the other cache policies against it, same box, alternating runs (tag = variant, untagged = the product = st17):
#      st0 (the old write-back default), st1 (sc0), st16 (sc1), st18 (sc1 nt), st19 (sc0 sc1 nt), ld2 / ld17 (policy bits on the loads),
#      recur (the recurrence integrator on top of st17) — Quadrotor2D two rounds each; the other three shipped tasks for st0 / st16 / st19;
#   2. st17 vs st0 where the working set streams from HBM (262 144 / 1 M / 4 M envs);
#   3. the tests written this session (safe_explorer_ppo controller, SAC normalisers, grouped env ids, Adam layouts, cuda legs of the
#      reference-pinned fixtures) and the env parity files on the new default.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s82; mkdir -p $O
for tag in st0 st1 st16 st18 st19 ld2 ld17; do
  timeout 200 python tools/ab_variant.py run $tag --tasks quadrotor_2D_track --rounds 2 --no-gate 2>&1 | tee $O/ab_$tag.log | grep tag= | cut -c1-200
done
timeout 200 python tools/ab_variant.py run recur --tasks quadrotor_2D_track --rounds 2 2>&1 | tee $O/ab_recur.log | grep 'tag=\|one-step' | cut -c1-250
for tag in st0 st16 st19; do
  timeout 300 python tools/ab_variant.py run $tag --tasks cartpole_stab,quadrotor_3D_track,quadrotor_3D_track_disturbed --rounds 1 --no-gate 2>&1 | tee $O/ab3_$tag.log | grep tag= | cut -c1-200
done
timeout 400 python tools/ab_variant.py run st0 --tasks quadrotor_2D_track --rounds 1 --envs 262144,1048576,4194304 --no-gate 2>&1 | tee $O/ab_sweep_st0.log | grep tag= | cut -c1-200
( time timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_dropin.py tests/test_gpu_sac_fused.py::test_adam_state_travels_between_the_fused_and_the_torch_layout \
    tests/test_ppo_cpu.py tests/test_sac_cpu.py tests/test_adversarial_golden.py tests/test_normalizers_golden.py tests/test_metrics_golden.py \
    tests/test_gpu_env_parity.py tests/test_gpu_parity_scale.py ) > $O/tests.log 2>&1; tail -15 $O/tests.log | cut -c1-300
