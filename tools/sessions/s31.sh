#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s31; mkdir -p $O
for N in 65536 131072 262144 1048576 4194304; do
  S=5000; [ $N -gt 1000000 ] && S=600
  python bench.py --envs $N --steps $S --warmup 200 --no-secondary --ppo-seeds 0 --no-cpu-baseline > $O/bench_q2track_$N.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/bench_q2track_$N.json').read().strip().split(chr(10))[-1]); print($N, round(d['roofline']['avg_launch_us'],2), '%.3e' % d['value'], round(d['roofline']['frac'],3))"
done
