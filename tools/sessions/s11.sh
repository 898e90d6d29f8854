#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s11; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_rollout_policy.py -m gpu -q ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log
