#!/bin/bash
# round 3, session 23: scg_ppo_grad variants — 16-byte-read tile transposes, in-place transposes for dW2 (8 instead of 20),
# ring-ordered dW2 staging
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s59; mkdir -p $O
run() {  # name, flags
  SCG_LEARN_FLAGS="$2" python -c "
from safe_control_gym_amd import _learn; _learn.build(12,128,2,'tanh',force=True)" || return
  python tools/learn_cost.py > $O/cost_$1.txt 2>&1; echo "== $1 ($2)"; grep -E "65536|262144|per tile" $O/cost_$1.txt
}
run base "-DSCG_TR_OLD -DSCG_DW2_V=0 -DSCG_STG_OLD"
run tr "-DSCG_DW2_V=0 -DSCG_STG_OLD"
run tr_ring "-DSCG_DW2_V=0"
run oldtr_v1 "-DSCG_TR_OLD -DSCG_DW2_V=1"
run v2 "-DSCG_DW2_V=2"
run v1 ""
timeout 600 python -m pytest tests/test_gpu_learn.py tests/test_gpu_sac_fused.py -x -q 2>&1 | tail -3
