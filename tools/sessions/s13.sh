#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s13; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 5000 $O/bench_driver.json; tail -5 $O/bench_driver.err
