#!/bin/bash
# round 3, session 12: final check — smoke, whole GPU suite, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s45; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
