#!/bin/bash
# round 5, GPU session 8: learner micro-changes (16-byte W2 image fill, 16-byte partial-sum loads): bitwise / fixture tests of everything that fills an MLP image,
# then the cost model of scg_ppo_grad and the steady-state PPO iteration.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s113; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_learn.py tests/test_gpu_sac_fused.py tests/test_gpu_rollout_policy.py -x -q 2>&1 | tail -5 | tee $O/pytest_learn.txt
timeout 200 python tools/learn_cost.py 2>&1 | grep -v amdgpu.ids | tee $O/learn_cost.txt
timeout 200 python tools/ppo_iter_times.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/ppo_iter_times.txt
