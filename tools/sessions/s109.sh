#!/bin/bash
# round 5, GPU session 4: the driver-style bench with every leg (stdout / stderr / exit code kept), the float32 closed-loop parity margins (64 and 4096 initial states).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s109; mkdir -p $O
rocm-smi --showclocks --showpower > $O/rocm_smi_before_bench.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee $O/bench.rc
rocm-smi --showclocks --showpower > $O/rocm_smi_after_bench.txt 2>&1
tail -25 $O/bench.err | cut -c1-300
python - <<'PY'
import json, os
txt = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s109/bench.json').read().strip()
if not txt:
    raise SystemExit('bench printed nothing')
d = json.loads(txt.splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['frac_by_clock'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('frac'), (v.get('chain_latency') or {}).get('frac_of_launch')) for k, v in d.get('secondary', {}).items()})
print('f64', d.get('f64', {}).get('avg_launch_us'), d.get('f64', {}).get('frac'))
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'wall_clock_to_two_consecutive_s', 'cold_start_s', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'), r.get('param_randomised', {}).get('median_s'))
print('multi_gpu', d.get('multi_gpu', {}).get('allreduce_us'))
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
timeout 400 python -m pytest tests/test_gpu_parity_scale.py -x -q -k closed_loop 2>&1 | tail -4 | tee $O/pytest_parity.txt
