#!/bin/bash
# round 3, session 40: the fused PPO / SAC kernels directly against the reference's PPOAgent.update / SACAgent.update fixtures
# (learner.npz + the hyper-parameter corners of learner_variants.npz)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s76; mkdir -p $O
timeout 250 python -m pytest tests/test_gpu_learn.py tests/test_gpu_sac_fused.py -q -k "reproduces_the_reference" 2>&1 | tee $O/log.txt | grep -E "passed|failed|^(FAILED|ERROR)|^E  +(AssertionError|assert|Mismatch|Max abs|Max rel|.*ppo/|.*sac/)" | cut -c1-300 | head -40
