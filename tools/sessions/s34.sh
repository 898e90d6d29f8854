#!/bin/bash
# round 3, session 2: new bench legs (PPO at 65 536 envs with randomised-init evaluation + measured target, SAC leg, launcher), new tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s34; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
timeout 1200 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rollout_policy.py tests/test_bullet_convergence.py tests/test_gpu_dropin.py -x -q -m gpu > $O/pytest.log 2>&1
tail -5 $O/pytest.log
