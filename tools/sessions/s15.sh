#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s15; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_rl.py -m gpu -q -x -k "rarl or rap" ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -30
timeout 300 python tools/sac_time_to_reward.py --budget 90 > $O/sac.json 2> $O/sac.err; tail -c 2500 $O/sac.json; tail -5 $O/sac.err
