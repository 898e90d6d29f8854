#!/bin/bash
# round 3, session 17: is anything slower than round 2's kernels?  Round-2 sources (commit 8affb10) compiled as variant "r02", same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s50; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 --task quadrotor_2D_track"
for rep in 1 2; do
for N in 65536 1048576 4194304; do
  S=$((1600000000 / N)); [ $S -gt 20000 ] && S=20000
  G=$S; [ $G -gt 1000 ] && G=1000
  for V in "" r02; do
    SCG_SPEC_TAG=$V timeout 300 python bench.py $B --envs $N --steps $S --warmup $((S / 10)) --graph-len $G > $O/${N}_${V:-r03}_$rep.json 2>> $O/err.log
    python - $O/${N}_${V:-r03}_$rep.json ${V:-r03} <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print(sys.argv[2], d['config']['envs_per_gpu'], d['config']['kernel_build'][:6], 'us', round(d['roofline']['avg_launch_us'], 2), 'frac', round(d['roofline']['frac'], 3))
PY
  done
done
done
