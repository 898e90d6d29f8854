#!/bin/bash
# round 3, session 44 (the last 12 GPU-seconds): store ablations of the headline kernel — without the 16 constraint rows (-64 B per
# env-step), and without them and the observation (-112 B)
cd "$GRAFT_REPO_ROOT" || exit 1
B="--steps 4000 --warmup 300 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
for tag in nocval nocvobs; do
  SCG_SPEC_TAG=$tag python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tag=[$tag]', round(d['roofline']['avg_launch_us'],4), 'us')"
done
