#!/bin/bash
# round 6: smoke, the test added last, the driver's bench command with this round's profile files in place
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s132; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_step_launch.py -x -q -m gpu -k "rng_layout or thresholds" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 200 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s132/bench_driver.json').read().strip().splitlines()[-1])
r = d['roofline']
print(json.dumps({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype', 'scaling', 'vs_baseline')}))
print({k: r[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel')})
print(json.dumps(r['frac_by_clock']))
print(json.dumps(r['learners']))
print(json.dumps(d['ppo']['iteration_ms']))
print(json.dumps({k: v for k, v in d['sac']['roofline'].items() if k != 'what'}))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['ppo']['median_s'], d['ppo']['median_s_partial_epochs'], d['sac']['median_s'], d['multi_gpu']['allreduce_us'])
PY
