#!/bin/bash
# round 3, session 9: where scg_ppo_grad's fixed cost goes (in-kernel timestamps), learner scheduler variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s42; mkdir -p $O
python tools/learn_cost.py > $O/cost_default.txt 2>&1; tail -8 $O/cost_default.txt
SCG_LEARN_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp" python -c "
from safe_control_gym_amd import _learn; print(_learn.build(12,128,2,'tanh',force=True))"
python tools/learn_cost.py > $O/cost_ilp.txt 2>&1; tail -8 $O/cost_ilp.txt
SCG_LEARN_FLAGS="-DSCG_L_TIMING" python -c "
from safe_control_gym_amd import _learn; print(_learn.build(12,128,2,'tanh',force=True))"
python tools/learn_cost.py --timeline > $O/cost_timeline.txt 2>&1; tail -14 $O/cost_timeline.txt
