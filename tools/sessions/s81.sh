#!/bin/bash
# round 4, first GPU session (≈ 11 GPU-minutes; run tools/sessions/s81_prepare.sh HERE first — it builds the tagged variants):
#   1. the whole GPU suite on the tree round 3 ended with (its last commits changed only host code and were verified on the CPU:
#      facade members, EnvSpec's error behaviour, seed validation) — incl. the three non-strict xfail cases of tests/test_gpu_dropin.py that
#      run the reference's examples / test matrix on the HIP handle (XPASS expected: then drop the xfail marks);
#   2. smoke() and the driver-style default bench;
#   3. the reference's example test matrix on the HIP handle, verbatim output kept;
#   4. same-box A/B of the round-3 candidates (tools/candidates/README.md): wide SoA row stores (ceiling ≈ -0.6 µs on the headline, ≈ -1.6 µs
#      on Quadrotor3D), write-through stores, the recurrence integrator (-1.4 % measured), wide + recurrence together; each with the float32
#      one-step gate of tools/ab_variant.py.  Ship a winner only after tests/test_gpu_env_parity.py + the f32 cases of test_gpu_config_fuzz.py.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s81; mkdir -p $O
( time timeout 480 python -m pytest tests -m gpu -q -rxX ) > $O/suite.log 2>&1; tail -8 $O/suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s81/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k in ('ppo', 'sac'):
    r = d.get(k, {})
    print(k, {q: r.get(q) for q in ('median_s', 'reached_two_consecutive', 'error')})
PY
( time timeout 300 python tools/run_reference_example.py matrix ) > $O/matrix_hip.log 2>&1; tail -4 $O/matrix_hip.log | cut -c1-300
( time timeout 200 python tools/run_reference_example.py train --algo ppo --system cartpole --task stab --env-steps 800 ) > $O/train_hip.log 2>&1; grep 'TRAINED\|Training done\|rror' $O/train_hip.log | tail -3 | cut -c1-300
for tag in wide wide2 st17 recur; do
  E=65536; case $tag in wide*) E=65536,262144;; esac          # the LDS footprint of the wide-row variants matters from 4 workgroups per CU on
  timeout 240 python tools/ab_variant.py run $tag --tasks quadrotor_2D_track,quadrotor_3D_track --rounds 2 --envs $E 2>&1 | tee $O/ab_$tag.log | tail -22 | cut -c1-250
done
timeout 100 python tools/ab_variant.py run widerecur --tasks quadrotor_2D_track --rounds 2 2>&1 | tee $O/ab_widerecur.log | tail -8 | cut -c1-250
