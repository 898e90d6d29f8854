#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s121
timeout 300 python -m pytest tests/test_gpu_learn.py -x -q 2>&1 | tail -4 | tee gpurun_out/s121/pytest.txt
