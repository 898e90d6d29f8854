#!/bin/bash
# round 3, session 42: the whole GPU suite, smoke() and the driver-style default bench on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s78; mkdir -p $O
( time timeout 260 python -m pytest tests -m gpu -x -q ) > $O/suite.log 2>&1; tail -4 $O/suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s78/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('target_return', 'wall_clock_to_two_consecutive_s', 'median_s', 'reached_two_consecutive', 'error')})
print('ppo16k', {q: d['ppo'].get('envs_16384', {}).get(q) for q in ('wall_clock_to_two_consecutive_s', 'median_s')})
print('sac grad steps', d['sac'].get('gradient_steps'), d['sac'].get('env_steps_per_s_incl_learning'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
