#!/bin/bash
# round 4, last GPU session (≈ 9 GPU-minutes): speculative reset draws now CartPole-only (s98: Quadrotor2D 4.19 -> 4.02 us without them) —
# the whole GPU suite, smoke, the rocprofv3 / PMC passes of the env kernels on these sources, then the driver-style bench.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s99; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -rxXs ) > $O/suite.log 2>&1; tail -6 $O/suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time SCG_PROFILE_ENV_ONLY=1 bash tools/profile_round4.sh ) > $O/profile.log 2>&1; tail -3 $O/profile.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s99/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'traffic', d['roofline']['traffic'])
print('f64', d.get('f64', {}).get('avg_launch_us'), 'secondary', {k: v.get('avg_launch_us') for k, v in d.get('secondary', {}).items()})
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)}, 'fused_rollout', d.get('fused_rollout', {}).get('ms_per_rollout'))
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'reached_two_consecutive', 'wall_clock_to_two_consecutive_s', 'error')}, r.get('envs_16384', {}).get('median_s'),
          r.get('full_epochs', {}).get('median_s'), r.get('param_randomised', {}).get('median_s'))
PY
