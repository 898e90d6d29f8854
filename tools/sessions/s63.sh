#!/bin/bash
# round 3, session 27: ppo_grad_kernel — tile inputs requested a tile ahead (default build of that moment) and weight operands of the
# data gradient read a block ahead (-DSCG_L_PIPE): neither moved the kernel (33.6 / 33.8 vs 33.9 us per tile), the code was not kept
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s63; mkdir -p $O
run() {
  SCG_LEARN_FLAGS="$2" python -c "
from safe_control_gym_amd import _learn; _learn.build(12,128,2,'tanh',force=True)" || return
  python tools/learn_cost.py $3 > $O/cost_$1.txt 2>&1; echo "== $1 ($2)"; grep -E "65536|262144|per tile|tile:|tile total" $O/cost_$1.txt
}
run ahead ""
run pipe "-DSCG_L_PIPE"
run ahead_t "-DSCG_L_TIMING" --timeline
run pipe_t "-DSCG_L_TIMING -DSCG_L_PIPE" --timeline
python -c "
from safe_control_gym_amd import _learn; _learn.build(12,128,2,'tanh',force=True)"
timeout 900 python -m pytest tests/test_gpu_learn.py -x -q 2>&1 | tail -3
