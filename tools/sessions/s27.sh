#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s27; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -4
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -6
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
python -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().split('\n')[-1])
print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['f64']['frac'], d['sequence']['collector_outputs']['env_steps_per_s'], d['sequence']['all_outputs']['frac'], d['fused_rollout']['env_steps_per_s'], d['ppo']['wall_clock_to_target_s'], d['cpu_baseline']['value'])"
tail -3 $O/bench_driver.err
