#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s17; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sequence.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -25
python - > $O/seq.log 2>&1 <<'PY'
import torch, json, bench
torch.cuda.set_device(0)
for task in ('quadrotor_2D_track', 'cartpole_stab', 'quadrotor_3D_track'):
    for K in (8, 32, 128):
        r = bench.sequence_leg(torch, 65536, K, task)
        print(task, K, round(r['us_per_control_step'], 3), '%.3e' % r['env_steps_per_s'], round(r['frac'], 3), r['finite_outputs'])
PY
cat $O/seq.log
