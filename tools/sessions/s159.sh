#!/bin/bash
# round 6 (second session), final tree (operand-exchange dW2): the whole GPU suite, learner profiles (copied into profiles/ on the box so that bench.py quotes
# them), the default and the driver-style bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s159; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_full.txt 2>&1; tail -6 $O/pytest_full.txt
bash tools/profile_round6.sh > $O/profile6.log 2>&1; tail -4 $O/profile6.log | cut -c1-400
cp gpurun_out/prof6/r06_kernel_stats_ppo_iteration.csv gpurun_out/prof6/r06_kernel_stats_sac_iteration.csv gpurun_out/prof6/r06_learner_kernel_sums.json profiles/
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.err
python - <<'PY'
import json
for f in ('bench_default', 'bench_driver'):
    d = json.loads(open(f'gpurun_out/s159/{f}.json').read().strip().splitlines()[-1])
    print(f, json.dumps({k: d[k] for k in ('value', 'ms_per_step')}), d['roofline']['frac'], d['roofline']['traffic'])
    print(' ', json.dumps(d['roofline'].get('learners')))
    print(' ', d.get('ppo', {}).get('wall_clock_to_two_consecutive_s'), d.get('ppo', {}).get('iterations'), d.get('ppo', {}).get('error'))
    print(' ', d.get('sac', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('param_randomised', {}).get('wall_clock_to_two_consecutive_s'), d.get('sac', {}).get('error'))
PY
