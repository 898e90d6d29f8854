#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s53; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -m gpu > $O/pytest.log 2>&1
tail -15 $O/pytest.log
