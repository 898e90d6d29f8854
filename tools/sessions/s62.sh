#!/bin/bash
# round 3, session 26: per-phase timeline of one warm ppo_grad tile, stamps kept in registers
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s62; mkdir -p $O
SCG_LEARN_FLAGS="-DSCG_L_TIMING" python -c "
from safe_control_gym_amd import _learn; _learn.build(12,128,2,'tanh',force=True)"
python tools/learn_cost.py --timeline > $O/timeline.txt 2>&1; tail -17 $O/timeline.txt
