#!/bin/bash
# round 3, session 31: wide-tile SAC kernels as the only path — SAC tests (fixture, eager, two ranks), cost, one seed of the bench leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s67; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_multirank.py tests/test_gpu_dropin.py -x -q -k "sac or SAC" 2>&1 | tail -3
python tools/sac_update_cost.py > $O/cost.txt 2>&1; tail -1 $O/cost.txt
python - <<'PY' 2>&1 | tail -12
import json, torch, bench
torch.cuda.set_device(0)
r = bench.sac_leg(torch, [1], 40.0)
print(json.dumps({k: r[k] for k in ('wall_clock_to_first_hit_s', 'wall_clock_to_two_consecutive_s', 'env_steps', 'gradient_steps', 'env_steps_per_s_incl_learning', 'best_eval_return')}))
PY
