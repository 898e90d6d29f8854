#!/bin/bash
# round 3, session 11: activations pipelined under the MFMA chains + prefetched tile inputs: learner cost, rollout kernel, SAC step, parity
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s44; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rollout_policy.py tests/test_gpu_sac_fused.py tests/test_gpu_rl.py -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python tools/learn_cost.py > $O/cost_new.txt 2>&1; tail -7 $O/cost_new.txt
python tools/sac_update_cost.py > $O/sac_cost.json 2>/dev/null; cat $O/sac_cost.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ppo_iteration -o p -- python tools/ppo_profile.py --fused-rollout --iters 20 --minibatch 65024 > $O/ppo_iteration.log 2>&1 < /dev/null
grep -h "^{" $O/ppo_iteration.log
f=$(find $O/ppo_iteration -name "*kernel_stats.csv" | head -1); grep "rollout_policy\|ppo_grad\|mlp_forward" $f | awk -F'",' '{print substr($1,1,60), $2}'
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
# serial-activation variant of the learner, same box
SCG_LEARN_FLAGS="-DSCG_MLP_SERIAL_ACT" python -c "
from safe_control_gym_amd import _learn; print(_learn.build(12,128,2,'tanh',force=True))"
python tools/learn_cost.py > $O/cost_serial_act.txt 2>&1; tail -7 $O/cost_serial_act.txt
