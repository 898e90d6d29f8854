#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s21; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rollout_policy.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -12
python tools/learn_cost.py 2>&1 | tail -8
