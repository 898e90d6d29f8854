#!/bin/bash
# round 3, session 14: N-sweep of the step kernels (2^16 ... 2^24 envs), Quadrotor2D-track and CartPole-stab
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s47; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
for T in quadrotor_2D_track cartpole_stab; do
for N in 65536 131072 262144 1048576 4194304 16777216; do
  S=$((2000000000 / N)); [ $S -gt 20000 ] && S=20000; [ $S -lt 100 ] && S=100
  W=$((S / 10)); G=$S; [ $G -gt 1000 ] && G=1000
  timeout 300 python bench.py --task $T --envs $N --steps $S --warmup $W --graph-len $G $B > $O/bench_${T}_$N.json 2>> $O/err.log
  python - $O/bench_${T}_$N.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    print(d['config']['workload'][:40], d['config']['envs_per_gpu'], 'us', round(d['roofline']['avg_launch_us'], 2), 'env-steps/s %.3e' % d['value'], 'frac', round(d['roofline']['frac'], 3))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
done
done
