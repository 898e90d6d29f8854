#!/bin/bash
# round 5: the driver-style bench of the final tree (PPO leg on 3 partial epochs x 16 minibatches)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s120; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee $O/bench.rc
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s120/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline']['frac_by_clock'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('frac')) for k, v in d.get('secondary', {}).items()})
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac'), v.get('traffic_bytes_per_env_step')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'wall_clock_to_two_consecutive_s', 'iterations', 'hyper', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'), r.get('param_randomised', {}).get('median_s'))
PY
