#!/bin/bash
# round 4, fifth GPU session (≈ 10 GPU-minutes): the SAC bookkeeping fold is REVERTED (s84: reduce_kernel<28,1> 7.7 -> 64.8 us), the
# learner's partial vectors leave as write-through 16-byte stores (A/B here), bench.py carries the PMC traffic of s84 and the SAC leg
# with flyable parameter randomisation.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s85; mkdir -p $O
( time timeout 600 python -m pytest -m gpu -q tests/test_gpu_learn.py tests/test_gpu_sac_fused.py tests/test_gpu_rl.py tests/test_gpu_multirank.py \
    tests/test_gpu_adversarial.py tests/test_gpu_rollout_policy.py ) > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
for tag in part0 "" part0 ""; do
  echo "== learn_cost SCG_LEARN_TAG=[$tag]"; SCG_LEARN_TAG=$tag timeout 120 python tools/learn_cost.py 2>&1 | grep 'minibatch\|per tile\|adam' | cut -c1-200
done | tee $O/learn_cost_ab.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s85/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'traffic', d['roofline']['traffic'], d['roofline']['valu_issue'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('valu_issue')) for k, v in d.get('secondary', {}).items()})
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac'), v.get('traffic_bytes_per_env_step')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'reached_two_consecutive', 'error')}, r.get('envs_16384', {}).get('median_s'), r.get('full_epochs', {}).get('median_s'))
pr = d.get('sac', {}).get('param_randomised', {})
print('sac param_randomised', {q: pr.get(q) for q in ('target_return', 'median_s', 'reached_two_consecutive', 'wall_clock_to_two_consecutive_s', 'best_eval_return', 'error')})
PY
timeout 120 python tools/timeline.py run 65536 > $O/timeline.txt 2>&1; tail -12 $O/timeline.txt
