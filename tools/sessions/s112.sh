#!/bin/bash
# round 5, GPU session 7 (final kernels: Quadrotor3D on max-ilp): HBM mix ceilings, the whole GPU suite, smoke, the driver-style bench (all legs),
# the rocprofv3 passes behind profiles/r05_* (re-taken on these sources: the PMC file carries their hash).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s112; mkdir -p $O
tools/hbm_mix 2>&1 | tee $O/hbm_mix.txt
( time timeout 900 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_full.txt 2>&1; tail -6 $O/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2 | tee $O/smoke.txt
rocm-smi --showclocks --showpower > $O/rocm_smi_before_bench.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee $O/bench.rc
rocm-smi --showclocks --showpower > $O/rocm_smi_after_bench.txt 2>&1
python - <<'PY'
import json, os
txt = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s112/bench.json').read().strip()
d = json.loads(txt.splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['frac_by_clock'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('frac'), (v.get('chain_latency') or {}).get('frac_of_launch')) for k, v in d.get('secondary', {}).items()})
print('f64', d.get('f64', {}).get('avg_launch_us'), d.get('f64', {}).get('frac'))
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'wall_clock_to_two_consecutive_s', 'iterations', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'), r.get('param_randomised', {}).get('median_s'))
print('multi_gpu', d.get('multi_gpu', {}).get('allreduce_us'))
PY
SCG_PROFILE_LEARNERS=1 bash tools/profile_round5.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
