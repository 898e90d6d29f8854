#!/bin/bash
# round 3, session 32: one seed of the bench SAC leg on the wide-tile kernels
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' 2>&1 | tail -4
import json, torch, bench
torch.cuda.set_device(0)
r = bench.sac_leg(torch, 1, 40.0)
print(json.dumps({k: r[k] for k in ('wall_clock_to_first_hit_s', 'wall_clock_to_two_consecutive_s', 'env_steps', 'gradient_steps', 'env_steps_per_s_incl_learning', 'best_eval_return')}))
PY
