#!/bin/bash
# round 3, session 41: the reference-generated fuzz rollouts (tests/golden/rollout_fuzz_*.npz) through the GPU parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s77; mkdir -p $O
timeout 280 python -m pytest tests/test_gpu_env_parity.py -q -k "fuzz" 2>&1 | tee $O/log.txt | grep -E "passed|failed|^(FAILED|ERROR)|^E  +(AssertionError|assert|Mismatch|Max abs|Max rel|fuzz)" | cut -c1-300 | head -50
