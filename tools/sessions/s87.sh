#!/bin/bash
# round 4, seventh GPU session (≈ 7 GPU-minutes): hyper-parameter probes of the two wall-clock legs under bench.py's protocol, 3 seeds each.
#   SAC (config #5 env, randomisation off): the actor kernels of the fused step use 128 of the 256 CUs at batch 4096 (128 tiles of 32 samples);
#       batch 8192 fills the chip — does the bigger batch buy fewer gradient steps?
#   PPO (65 536 envs): optimiser steps per iteration around the shipped 2 x 32 x 16 256.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s87; mkdir -p $O
sac() { tag=$1; shift; timeout 240 python - "$@" > $O/sac_$tag.json 2> $O/sac_$tag.err <<'PY'
import json, sys, torch, bench
kw = dict(a.split('=') for a in sys.argv[1:])
kw = {k: (float(v) if '.' in v or 'e' in v else int(v)) for k, v in kw.items()}
torch.cuda.set_device(0)
r = bench.sac_leg(torch, 3, 30.0, **kw)
print(json.dumps({k: r[k] for k in ('wall_clock_to_two_consecutive_s', 'gradient_steps', 'env_steps', 'median_s', 'best_eval_return', 'target_return')}))
PY
  echo "sac $tag $(tail -1 $O/sac_$tag.json | cut -c1-400)"; }
sac b4096_u16 batch=4096 updates_per_step=16
sac b8192_u16 batch=8192 updates_per_step=16
sac b8192_u8 batch=8192 updates_per_step=8
sac b8192_u12_lr2 batch=8192 updates_per_step=12 lr=2e-3
sac b2048_u32 batch=2048 updates_per_step=32
run() { tag=$1; shift; timeout 200 python tools/ppo_seeds.py --envs 65536 --seeds 3 --budget 6 "$@" > $O/ppo_$tag.json 2> $O/ppo_$tag.err
  python - $O/ppo_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    print('ppo', sys.argv[2], 'two_consec', [round(x, 2) if x else None for x in d['wall_clock_to_two_consecutive_s']], 'its', d['iterations'], 'median', d['median_s'])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run mb16256_e2_cap24 --minibatch 16256 --epochs 2 --mb-per-epoch 24
run mb16256_e2_cap40 --minibatch 16256 --epochs 2 --mb-per-epoch 40
run mb16256_e2_cap32_lr3 --minibatch 16256 --epochs 2 --mb-per-epoch 32 --lr 3e-3
run mb8128_e2_cap64 --minibatch 8128 --epochs 2 --mb-per-epoch 64
