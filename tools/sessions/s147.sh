#!/bin/bash
# round 6 (second session): new test cases — SAC batches above the 512-workgroup cap (multi-tile loops of the wide kernels), PPO gradients vs autograd with
# several tiles per wave
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s147; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_learn.py tests/test_gpu_sac_fused.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
