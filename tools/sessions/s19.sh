#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s19; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -30
