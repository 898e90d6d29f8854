#!/bin/bash
# round 3, session 30: wide-tile SAC kernels (the NT waves of a workgroup share one 32-sample tile) — correctness per launch, cost
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s66; mkdir -p $O
t() { SCG_SAC_WIDE=$1 timeout 300 python -m pytest tests/test_gpu_sac_fused.py -x -q > $O/test_$1.log 2>&1; echo "mask $1: $(tail -1 $O/test_$1.log)"; }
t 63
if ! grep -q " passed" $O/test_63.log || grep -q failed $O/test_63.log; then
  grep -E "^E " $O/test_63.log | head -8
  for m in 1 8 16 2 32 4; do t $m; done
fi
SCG_SAC_WIDE=63 python tools/sac_update_cost.py > $O/cost_wide.txt 2>&1; tail -1 $O/cost_wide.txt
