#!/bin/bash
# round 4, third GPU session (≈ 14 GPU-minutes): the product now ships write-through stores (SCG_ST_AUX=17) AND the recurrence integrator
# (SCG_Q2_RECUR) — s82 measured 5.21 (write-back) -> 4.39 (write-through) -> 4.17 us (+ recurrence) on the headline.
#   1. the WHOLE GPU suite on that tree (no -x: s82's run stopped at its first failure, a test-side assumption since fixed);
#   2. smoke + the driver-style default bench;
#   3. workgroup size under the new store policy (64 threads was chosen under write-back stores): 128 / 256;
#   4. PPO at 65 536 envs with partial epochs (extra['minibatches_per_epoch']): optimiser steps per iteration as at 16 384 envs.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s83; mkdir -p $O
( time timeout 700 python -m pytest tests -m gpu -q -rxXs ) > $O/suite.log 2>&1; tail -12 $O/suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s83/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('f64', d.get('f64', {}).get('avg_launch_us'), 'secondary', {k: v.get('avg_launch_us') for k, v in d.get('secondary', {}).items()})
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'reached_two_consecutive', 'error')}, r.get('envs_16384', {}).get('median_s'))
PY
for tag in blk128 blk256; do
  timeout 300 python tools/ab_variant.py run $tag --tasks quadrotor_2D_track,cartpole_stab,quadrotor_3D_track --rounds 1 --no-gate 2>&1 | tee $O/ab_$tag.log | grep tag= | cut -c1-200
done
run() { tag=$1; shift; timeout 200 python tools/ppo_seeds.py --envs 65536 --seeds 3 --budget 6 "$@" > $O/ppo_$tag.json 2> $O/ppo_$tag.err
  python - $O/ppo_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    print(sys.argv[2], 'two_consec', [round(x, 2) if x else None for x in d['wall_clock_to_two_consecutive_s']], 'its', d['iterations'], 'median', d['median_s'])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run mb16256_e2_cap32 --minibatch 16256 --epochs 2 --mb-per-epoch 32
run mb16256_e1_cap64 --minibatch 16256 --epochs 1 --mb-per-epoch 64
run mb32512_e2_cap16 --minibatch 32512 --epochs 2 --mb-per-epoch 16
run mb32512_e2_cap32 --minibatch 32512 --epochs 2 --mb-per-epoch 32
run mb65024_e2_cap16 --minibatch 65024 --epochs 2 --mb-per-epoch 16
