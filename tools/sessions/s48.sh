#!/bin/bash
# round 3, session 15: the streaming-regime build (default scheduler) at 4 M / 16 M envs
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s47; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
for T in quadrotor_2D_track; do
for N in 1048576 2097152 4194304 16777216; do
  S=$((2000000000 / N)); [ $S -lt 100 ] && S=100
  timeout 300 python bench.py --task $T --envs $N --steps $S --warmup $((S / 10)) --graph-len $S $B > $O/bench_${T}_$N.json 2>> $O/err.log
  python - $O/bench_${T}_$N.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print(d['config']['envs_per_gpu'], 'us', round(d['roofline']['avg_launch_us'], 2), 'env-steps/s %.3e' % d['value'], 'frac', round(d['roofline']['frac'], 3))
PY
done
done
