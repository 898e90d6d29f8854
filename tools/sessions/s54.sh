#!/bin/bash
# round 3, session 19: hyper-parameter probes of the 16 384-env PPO point under the round-3 protocol
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s54; mkdir -p $O
run() { tag=$1; shift; timeout 200 python tools/ppo_seeds.py --envs 16384 --seeds 4 --budget 6 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    print(sys.argv[2], 'two_consec', [round(x, 2) if x else None for x in d['wall_clock_to_two_consecutive_s']], 'its', d['iterations'], 'median', d['median_s'])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run mb32512_e4 --minibatch 32512 --epochs 4
run mb32512_e2 --minibatch 32512 --epochs 2
run mb16256_e2 --minibatch 16256 --epochs 2
run mb16256_e4 --minibatch 16256 --epochs 4
run mb32512_e3 --minibatch 32512 --epochs 3
run mb65024_e4 --minibatch 65024 --epochs 4
