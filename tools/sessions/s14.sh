#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s14; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_rl.py -m gpu -q -x -s -k "64_initial or rarl or rap" ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -40
python tools/ppo_seeds.py --envs 16384 --minibatch 65536 --seeds 6 --budget 10 > $O/ppo_16k.json 2> $O/ppo_16k.err; tail -c 1200 $O/ppo_16k.json; tail -3 $O/ppo_16k.err
python tools/ppo_seeds.py --envs 65536 --minibatch 262144 --seeds 6 --budget 15 > $O/ppo_64k_mb256k.json 2> $O/ppo_64k.err; tail -c 1200 $O/ppo_64k_mb256k.json; tail -3 $O/ppo_64k.err
python tools/ppo_seeds.py --envs 65536 --minibatch 65536 --seeds 6 --budget 15 > $O/ppo_64k_mb64k.json 2>> $O/ppo_64k.err; tail -c 1200 $O/ppo_64k_mb64k.json
