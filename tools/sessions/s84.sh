#!/bin/bash
# round 4, fourth GPU session (≈ 18 GPU-minutes), on the tree that ships: write-through stores in the one-launch-per-step kernels, WRITE-BACK
# stores in the K-steps-per-launch kernels (s83's bench showed scg_step_sequence 9-14 % slower when written through), the recurrence
# integrator, the SAC step's bookkeeping folded into the critics' reduction (8 launches), partial epochs as the 65 536-env PPO default.
#   1. the whole GPU suite (incl. the 8-rank gloo tests, the parity-margin record);  2. smoke + the driver-style default bench;
#   3. sequence kernels: write-back (shipped) vs write-through, same box;  4. the rocprofv3 passes behind profiles/r04_*;
#   5. per-phase shader-clock timeline of the shipped Quadrotor2D step kernel (LAST: it swaps a timing build in for the shipped library).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s84; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -rxXs ) > $O/suite.log 2>&1; tail -8 $O/suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s84/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('f64', d.get('f64', {}).get('avg_launch_us'), 'secondary', {k: v.get('avg_launch_us') for k, v in d.get('secondary', {}).items()})
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
print('fused_rollout', d.get('fused_rollout', {}).get('ms_per_rollout'))
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'reached_two_consecutive', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'))
PY
for tag in seq17 "" seq17 ""; do
SCG_SPEC_TAG=$tag python - <<'PY' 2>&1 | tail -1
import os, torch, bench
torch.cuda.set_device(0)
r = bench.sequence_leg(torch, 65536)
print('sequence tag=[%s]' % os.environ.get('SCG_SPEC_TAG', ''), {k: round(v['us_per_control_step'], 4) for k, v in r.items() if isinstance(v, dict)})
PY
done
( time bash tools/profile_round4.sh ) > $O/profile.log 2>&1; tail -5 $O/profile.log
timeout 120 python tools/timeline.py run 65536 > $O/timeline.txt 2>&1; cat $O/timeline.txt | tail -12
