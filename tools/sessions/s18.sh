#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s18; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_rl.py -m gpu -q -x -k "async" ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -15
python tools/ppo_seeds.py --envs 16384 --minibatch 65536 --seeds 6 --budget 10 > $O/ppo_16k.json 2> $O/ppo_16k.err; tail -c 1000 $O/ppo_16k.json; tail -3 $O/ppo_16k.err
python tools/ppo_seeds.py --envs 65536 --minibatch 65536 --seeds 6 --budget 15 > $O/ppo_64k.json 2> $O/ppo_64k.err; tail -c 1000 $O/ppo_64k.json; tail -3 $O/ppo_64k.err
python examples/train_ppo.py --seed 2 --quiet 2>&1 | tail -c 600
