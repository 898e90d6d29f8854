#!/bin/bash
# round 6 (second session): Adam arithmetic pinned (scg_adam.h); scg_sac_update_n == n x scg_sac_update; two waves per SIMD in the critic pass
# (mlp_forward_kernel, 8 waves behind one image) and in the policy rollout (32 envs per wave, 8 waves per workgroup at 65 536 envs): A/B inside one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s135; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_rollout_policy.py tests/test_gpu_multirank.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python tools/sac_step_ab.py shipped base > $O/sac_ab.txt 2>&1; tail -5 $O/sac_ab.txt
run() { # label, env assignments...
  L=$1; shift
  env "$@" timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update']['value_loss'])"
}
for rep in 1 2; do
  run "new(rollout 32x8, critic 8 waves)      " X=1
  run "rollout 64x4 (old geometry)            " SCG_ROLLOUT_EPW=64 SCG_ROLLOUT_WPW=4
  run "rollout 32x8, default scheduling       " SCG_SPEC_TAG=dsched
  run "old learn lib (critic 4 waves, reduce) " SCG_LEARN_TAG=base
  run "all old                                " SCG_LEARN_TAG=base SCG_ROLLOUT_EPW=64 SCG_ROLLOUT_WPW=4
done 2>&1 | tee $O/ppo_ab.txt
