#!/bin/bash
# round 3, session 38: reference vec-API info dicts on random configs; learner / SAC / policy-rollout kernels at network shapes no
# shipped task has (hidden 96, odd input widths, widest inputs); scg_gae at every dispatch boundary
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s74; mkdir -p $O
run() { timeout 280 python -m pytest "$@" -q 2>&1 | tee -a $O/log.txt | grep -E "passed|failed|^(FAILED|ERROR)|^E  +(AssertionError|assert|Mismatch|Max abs|.*seed=|.*T=)" | cut -c1-300 | head -40; }
run tests/test_gpu_config_fuzz.py -k vec_api
run tests/test_gpu_gae.py
run tests/test_gpu_learn.py
run tests/test_gpu_sac_fused.py
run tests/test_gpu_rollout_policy.py -k "96"
