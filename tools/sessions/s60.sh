#!/bin/bash
# round 3, session 24: where one warm tile of ppo_grad_kernel spends its time (in-kernel timestamps per phase)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s60; mkdir -p $O
for V in "-DSCG_TR_OLD -DSCG_DW2_V=0" "-DSCG_DW2_V=1"; do
SCG_LEARN_FLAGS="-DSCG_L_TIMING $V" python -c "
from safe_control_gym_amd import _learn; _learn.build(12,128,2,'tanh',force=True)"
echo "== $V"; python tools/learn_cost.py --timeline 2>&1 | tail -17
done
