#!/bin/bash
# round 6 (second session): new tests — one-tile form == accumulating form bit for bit (scg_learn_force_accumulating_form), rollout geometries with 8 waves
# per workgroup on small batches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s145; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rollout_policy.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
