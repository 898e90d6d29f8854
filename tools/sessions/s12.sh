#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s12; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 6000 $O/bench_driver.json; tail -5 $O/bench_driver.err
( time python bench.py --no-cpu-baseline --ppo-seeds 0 ) > $O/bench_long.json 2> $O/bench_long.err
python -c "import json; d=json.loads(open('$O/bench_long.json').read().strip().split('\n')[-1]); print('long', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
bash tools/profile_round.sh > $O/profile.log 2>&1
tail -5 $O/profile.log
