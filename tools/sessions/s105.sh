cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s105; mkdir -p $O
T=quadrotor_2D_track
for N in 65536 131072 262144 1048576 4194304 16777216; do
  S=4000; [ $N -ge 1048576 ] && S=600; [ $N -ge 16777216 ] && S=100
  timeout 100 python bench.py --task $T --envs $N --steps $S --warmup $((S/8)) --graph-len $((S<1000?S:1000)) --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 > $O/bench_${T}_$N.json 2> $O/err.txt
  python - $O/bench_${T}_$N.json $T $N <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], 'us/launch %.3f' % d['roofline']['avg_launch_us'], 'env-steps/s %.3e' % d['value'], 'frac %.4f' % d['roofline']['frac'])
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'failed', e)
PY
done
