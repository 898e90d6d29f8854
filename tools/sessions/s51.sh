#!/bin/bash
# round 3, session 18: data-parallel fused SAC step (two ranks on one GPU, gloo) + the fused SAC tests after the phases split
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s51; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_multirank.py -q -m gpu > $O/pytest.log 2>&1
tail -25 $O/pytest.log
python tools/sac_update_cost.py 2>/dev/null | tail -1
