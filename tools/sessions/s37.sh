#!/bin/bash
# round 3, session 5: whole GPU suite, the default bench line, then the round's rocprofv3 passes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s37; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
tail -15 $O/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
bash tools/profile_round3.sh > $O/profile_round3.log 2>&1
tail -5 $O/profile_round3.log
