#!/bin/bash
# round 3, session 25: ppo_grad_kernel with the transposed data gradient (12 tile transposes instead of 28, none for dW1),
# hoisted loss scalars, C-input dW1 accumulation, ring-ordered dW2 staging: correctness, cost model, per-phase timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s61; mkdir -p $O
python -c "
from safe_control_gym_amd import _learn, _sac
for k in [(12,128,2,'tanh'),(6,64,2,'tanh'),(4,32,1,'leaky_relu'),(24,128,4,'tanh')]:
    try: _learn.build(*k)
    except Exception as e: print(k, e)
for k in [(24,128,4,'relu'),(6,32,2,'relu'),(12,64,2,'relu')]: _sac.build(*k)"
timeout 900 python -m pytest tests/test_gpu_learn.py tests/test_gpu_sac_fused.py -x -q 2>&1 | tail -5
python tools/learn_cost.py > $O/cost.txt 2>&1; tail -8 $O/cost.txt
SCG_LEARN_FLAGS="-DSCG_L_TIMING" python -c "
from safe_control_gym_amd import _learn; _learn.build(12,128,2,'tanh',force=True)"
python tools/learn_cost.py --timeline > $O/timeline.txt 2>&1; tail -17 $O/timeline.txt
