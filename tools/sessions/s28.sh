#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s28; mkdir -p $O
for i in 1 2; do
  ( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_$i.log 2>&1
  grep -v "^$" $O/pytest_$i.log | tail -3
done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['roofline']['frac'], {k: round(v, 4) if isinstance(v, float) else v for k, v in d['sequence'].items() if k != 'note'}, d['ppo']['wall_clock_to_target_s'])"
