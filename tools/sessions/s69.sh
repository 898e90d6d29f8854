#!/bin/bash
# round 3, session 33 (after the container was re-created): the wide-tile fused SAC step on a fresh build — SAC tests, cost per
# gradient step, rocprofv3 kernel statistics of a SAC iteration (replaces profiles/r03_kernel_stats_sac_iteration.csv and
# r03_sac_update_cost.json, which described the one-wave-per-tile kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s69; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_sac_fused.py -x -q 2>&1 | tail -2
python tools/sac_update_cost.py > $O/cost.json 2> $O/cost.err; tail -1 $O/cost.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sac_iteration -o p -- \
    python tools/sac_time_to_reward.py --budget 10 --eval-every 100000 > $O/sac_iteration.log 2>&1 < /dev/null
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*.db' -delete
f=$(find $O/sac_iteration -name '*kernel_stats.csv' | head -1); head -12 "$f" | cut -c1-160
tail -3 $O/sac_iteration.log | cut -c1-300
