#!/bin/bash
# round 3, session 3: fused SAC update (tests, cost per gradient step, time-to-reward), RARL / RAP collector pins
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_adversarial.py -q -m gpu > $O/pytest.log 2>&1
tail -30 $O/pytest.log
timeout 300 python tools/sac_update_cost.py > $O/sac_cost.json 2> $O/sac_cost.err; cat $O/sac_cost.json; tail -3 $O/sac_cost.err
timeout 400 python tools/sac_time_to_reward.py --budget 90 --eval-every 50 > $O/sac_ttr.json 2> $O/sac_ttr.err; tail -c 1500 $O/sac_ttr.json; tail -3 $O/sac_ttr.err
