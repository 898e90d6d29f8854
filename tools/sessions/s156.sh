#!/bin/bash
# round 6 (second session), final: the default and the driver-style bench lines with the final bench.py (plain-loop wall quoted beside the leg's)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s156; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 200 $O/bench_driver.err
python - <<'PY'
import json
for f in ('bench_default', 'bench_driver'):
    d = json.loads(open(f'gpurun_out/s156/{f}.json').read().strip().splitlines()[-1])
    print(f, json.dumps({k: d[k] for k in ('value', 'ms_per_step')}), d['roofline']['frac'], d['roofline']['traffic'])
    print(' ', json.dumps(d['roofline'].get('learners')))
    print(' ', d.get('ppo', {}).get('wall_clock_to_two_consecutive_s'), d.get('ppo', {}).get('iterations'), d.get('ppo', {}).get('error'))
    print(' ', d.get('sac', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('param_randomised', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('error'))
PY
