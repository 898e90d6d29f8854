#!/bin/bash
# round 3, session 22: staggered issue priority for the two waves of a SIMD in the 8-wave rollout geometry
cd "$GRAFT_REPO_ROOT" || exit 1
for V in nostag stag; do for G in 64 328; do
  SCG_SPEC_TAG=$V SCG_ROLLOUT_EPW=$G python - <<PY
import json, torch, bench
torch.cuda.set_device(0)
r = bench.fused_rollout_leg(torch, 65536)
print('$V geometry $G', round(r['ms_per_rollout'], 4), 'ms;', '%.3e' % r['env_steps_per_s'], 'env-steps/s')
PY
done; done
