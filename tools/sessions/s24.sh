#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s24; mkdir -p $O
python tools/ppo_profile.py --fused-rollout --iters 30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ppo -- python $GRAFT_REPO_ROOT/tools/ppo_profile.py --fused-rollout --iters 30 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-150
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
