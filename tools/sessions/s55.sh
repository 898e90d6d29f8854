#!/bin/bash
# round 3, session 20: SAC hyper-parameter probes (fused update), 2 seeds each, old fixed-init evaluation of tools/sac_time_to_reward.py
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s55; mkdir -p $O
run() { tag=$1; shift; timeout 200 python tools/sac_time_to_reward.py --budget 40 --eval-every 50 --seeds 2 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], [(round(r['wall_clock_to_target_s'], 1) if r['wall_clock_to_target_s'] else None, round(r['best_eval_return'], 1), r['vector_steps']) for r in d['runs']])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run u16_e2048_b4096 --updates-per-step 16 --envs 2048 --batch 4096
run u32_e2048_b4096 --updates-per-step 32 --envs 2048 --batch 4096
run u16_e1024_b4096 --updates-per-step 16 --envs 1024 --batch 4096
run u16_e2048_b8192 --updates-per-step 16 --envs 2048 --batch 8192
run u16_e2048_b4096_lr2 --updates-per-step 16 --envs 2048 --batch 4096 --lr 2e-3
run u8_e1024_b2048 --updates-per-step 8 --envs 1024 --batch 2048
run u16_e2048_b2048 --updates-per-step 16 --envs 2048 --batch 2048
