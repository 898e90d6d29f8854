#!/bin/bash
# round 3, session 29: fused SAC step with the sampling folded into the first actor launch and the Adam steps into the reductions
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s65; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_multirank.py -x -q -k "sac or SAC" 2>&1 | tail -3
python tools/sac_update_cost.py > $O/cost.txt 2>&1; tail -12 $O/cost.txt
