#!/bin/bash
# round 3, session 4: whole GPU suite on the rebuilt libraries (FLAG_GROUND, grouped env_configs, EPW geometries, 16-byte rows),
# then SAC hyper-parameter probes for the bench leg (fused update)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s36; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
for V in "8 4096 4096" "16 4096 4096" "4 4096 2048" "16 2048 4096"; do
  set -- $V
  timeout 120 python tools/sac_time_to_reward.py --budget 50 --eval-every 50 --updates-per-step $1 --envs $2 --batch $3 > $O/sac_u$1_e$2_b$3.json 2> $O/sac_u$1_e$2_b$3.err
  python - $O/sac_u$1_e$2_b$3.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d['runs'][0]
    print(sys.argv[1].split('/')[-1], 'target', round(d['target_return'], 1), 'reached_s', r['wall_clock_to_target_s'], 'best', round(r['best_eval_return'], 1), 'env_steps', r['env_steps'], 'vsteps', r['vector_steps'])
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
done
