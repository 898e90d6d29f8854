#!/bin/bash
# round 3, session 16: same-box A/B at 4 M envs (streaming regime): default scheduler (base), max-ilp, default without PreDraw
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s49; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 --task quadrotor_2D_track --envs 4194304 --steps 400 --warmup 40 --graph-len 400"
for rep in 1 2; do for V in base ilp nopre; do
  SCG_SPEC_TAG=$V timeout 300 python bench.py $B > $O/${V}_$rep.json 2>> $O/err.log
  python - $O/${V}_$rep.json $V <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print(sys.argv[2], d['config']['kernel_build'], 'us', round(d['roofline']['avg_launch_us'], 2), 'frac', round(d['roofline']['frac'], 3))
PY
done; done
