#!/bin/bash
# round 3, session 13: segmented GAE kernel — parity and timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s46; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gae.py tests/test_gpu_rl.py tests/test_gpu_adversarial.py tests/test_gpu_rollout_policy.py -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python - > $O/gae.json <<'PY'
import json, torch, bench
torch.cuda.set_device(0)
print(json.dumps(bench.gae_leg(torch)))
PY
cat $O/gae.json
