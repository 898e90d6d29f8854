#!/bin/bash
# round 4, really the last GPU session (≈ 3 GPU-minutes): scheduler strategies on the final sources (speculative draws off for the quadrotors), then
# the driver-style bench with the PMC file of these sources in place.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s100; mkdir -p $O
for v in "q2def quadrotor_2D_track" "q3maxilp quadrotor_3D_track" "q3ddef quadrotor_3D_track_disturbed" "q3dmaxilp quadrotor_3D_track_disturbed"; do
  set -- $v
  timeout 100 python tools/ab_variant.py run $1 --tasks $2 --rounds 1 --no-gate 2>&1 | tee $O/ab_$1.log | grep tag= | cut -c1-200
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s100/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'traffic', d['roofline']['traffic'], d['roofline']['valu_issue'])
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac'), v.get('traffic_bytes_per_env_step')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'wall_clock_to_two_consecutive_s', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'), r.get('param_randomised', {}).get('median_s'))
PY
