#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s23; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_learn.py tests/test_gpu_rl.py tests/test_gpu_rollout_policy.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -8
python tools/ppo_seeds.py --envs 16384 --minibatch 65536 --seeds 6 --budget 10 > $O/ppo_16k.json 2> $O/ppo_16k.err; tail -c 900 $O/ppo_16k.json; tail -3 $O/ppo_16k.err
python tools/ppo_seeds.py --envs 65536 --minibatch 65536 --seeds 6 --budget 15 > $O/ppo_64k.json 2> $O/ppo_64k.err; tail -c 900 $O/ppo_64k.json; tail -3 $O/ppo_64k.err
