#!/bin/bash
# round 3, session 6: A/B of the speculative reset draws inside the integrator (PreDraw) on the headline + CartPole kernels, then parity
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s38; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 --steps 20000 --warmup 2000"
for T in quadrotor_2D_track cartpole_stab; do
  for rep in 1 2; do
    python bench.py --task $T $B > $O/${T}_pre_$rep.json 2>> $O/err.log
    SCG_SPEC_TAG=nopre SCG_SPEC_FLAGS=-DSCG_NO_PREDRAW python bench.py --task $T $B > $O/${T}_nopre_$rep.json 2>> $O/err.log
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/s38/*.json'))):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][0])
        print(os.path.basename(f), 'period_us', round(d['roofline']['avg_launch_us'], 3), 'ms_per_step', round(d['ms_per_step'] * 1e3, 3), 'frac', d['roofline']['frac'])
    except Exception as e:
        print(f, 'failed', e)
PY
timeout 1500 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_sequence.py tests/test_gpu_rollout_policy.py tests/test_gpu_rl.py -q -m gpu > $O/pytest.log 2>&1
tail -6 $O/pytest.log
