#!/bin/bash
# round 3, session 21: 8-wave workgroups (two waves per SIMD) for the policy rollout at 65 536 envs
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s57; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rollout_policy.py -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for G in 64 328 32; do
  SCG_ROLLOUT_EPW=$G python - <<PY
import json, torch, bench
torch.cuda.set_device(0)
r = bench.fused_rollout_leg(torch, 65536)
print('geometry $G', round(r['ms_per_rollout'], 4), 'ms per 32-step collection incl. critic passes + GAE;', '%.3e' % r['env_steps_per_s'], 'env-steps/s')
PY
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for G in 64 328; do
SCG_ROLLOUT_EPW=$G timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$G -o p -- python -c "
import torch, bench
torch.cuda.set_device(0)
print(bench.fused_rollout_leg(torch, 65536))" > $O/prof_$G.log 2>&1 < /dev/null
f=$(find $O/prof_$G -name "*kernel_stats.csv" | head -1); echo "geometry $G:"; grep "rollout_policy\|mlp_forward" $f | awk -F'",' '{print substr($1,1,70), $2}'
done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
