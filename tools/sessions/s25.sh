#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s25; mkdir -p $O
python tools/ppo_seeds.py --envs 16384 --minibatch 65024 --seeds 6 --budget 10 > $O/ppo_16k.json 2> $O/ppo_16k.err; tail -c 700 $O/ppo_16k.json | head -c 500; tail -3 $O/ppo_16k.err
python tools/ppo_seeds.py --envs 65536 --minibatch 65024 --seeds 6 --budget 15 > $O/ppo_64k.json 2> $O/ppo_64k.err; tail -c 700 $O/ppo_64k.json | head -c 500; tail -3 $O/ppo_64k.err
