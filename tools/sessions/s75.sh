#!/bin/bash
# round 3, session 39: partial resets vs the oracle and random-state round trips on random configs; vec-API info dicts re-run
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s75; mkdir -p $O
timeout 280 python -m pytest tests/test_gpu_config_fuzz.py -q -k "partial_reset or rollout_random" 2>&1 | tee $O/log.txt | grep -E "passed|failed|^(FAILED|ERROR)|^E  +(AssertionError|assert|Mismatch|Max abs|.*seed=)" | cut -c1-300 | head -40
