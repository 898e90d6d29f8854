#!/bin/bash
# round 3, session 35: differential test over random task configs (generic f64 kernels vs the oracle, tests/test_gpu_config_fuzz.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s71; mkdir -p $O
timeout 280 python -m pytest tests/test_gpu_config_fuzz.py -q 2>&1 | tee $O/fuzz.log | grep -E "passed|failed|AssertionError: " | cut -c1-220 | head -60
