#!/bin/bash
# round 5, last GPU session: (product code = what s112 validated: 796 GPU tests) sequence / rollout PMC passes, N-sweep of the final kernels, PPO seed spread,
# the driver-style bench with the PMC file of these sources in place.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s115; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kstep_oracle.py tests/test_gpu_split_step.py tests/test_gpu_learn.py -x -q 2>&1 | tail -3 | tee $O/pytest_quick.txt
SCG_PROFILE_SPECS="quadrotor_2D_track:65536:f32" SCG_PROFILE_SEQ=1 bash tools/profile_round5.sh > $O/profile_seq.log 2>&1; tail -2 $O/profile_seq.log
mkdir -p $O/sweep
for task in quadrotor_2D_track cartpole_stab; do
  for n in 65536 131072 262144 1048576 4194304 16777216; do
    steps=$(( 2000000000 / n )); [ $steps -gt 4000 ] && steps=4000; [ $steps -lt 200 ] && steps=200
    python bench.py --task $task --envs $n --steps $steps --warmup $(( steps / 10 )) --graph-len $(( steps < 1000 ? steps : 1000 )) --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 2>/dev/null | tail -1 > $O/sweep/${task}_$n.json
    python -c "
import json; d=json.load(open('$O/sweep/${task}_$n.json')); r=d['roofline']; print('$task', $n, '%.3e env-steps/s' % d['value'], '%.2f us' % r['avg_launch_us'], 'frac %.3f' % r['frac'], r['kernel'][-60:])"
  done
done 2>&1 | tee $O/sweep.txt
timeout 300 python tools/ppo_seeds.py --envs 65536 --mb-per-epoch 32 --seeds 8 2>/dev/null | tail -1 > $O/ppo_seeds_8.json; python -c "
import json; d=json.load(open('$O/ppo_seeds_8.json')); print('ppo 8 seeds', d['wall_clock_to_two_consecutive_s'], d['iterations'], 'median', d['median_s'])"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee $O/bench.rc
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s115/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline']['frac_by_clock'])
print('valu_issue', d['roofline']['valu_issue'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('frac'), v.get('valu_issue', {}) and v['valu_issue'].get('frac'), (v.get('chain_latency') or {}).get('frac_of_launch')) for k, v in d.get('secondary', {}).items()})
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac'), v.get('traffic_bytes_per_env_step')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'wall_clock_to_two_consecutive_s', 'iterations', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'), r.get('param_randomised', {}).get('median_s'))
PY
