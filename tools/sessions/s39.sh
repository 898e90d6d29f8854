#!/bin/bash
# round 3, session 7: same-box A/B of the step-kernel variants (prebuilt in-tree: default = PreDraw on, nopre, ilp = max-ilp scheduler, ilpnopre)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s39; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 --steps 20000 --warmup 2000"
for rep in 1 2 3; do
for T in quadrotor_2D_track cartpole_stab; do
  python bench.py --task $T $B > $O/${T}_default_$rep.json 2>> $O/err.log
  for V in nopre ilp ilpnopre; do
    SCG_SPEC_TAG=$V python bench.py --task $T $B > $O/${T}_${V}_$rep.json 2>> $O/err.log
  done
done
done
python - <<'PY'
import json, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/s39/*.json'))):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][0])
        key = os.path.basename(f).rsplit('_', 1)[0]
        acc[key].append((round(d['roofline']['avg_launch_us'], 3), d['config']['kernel_build']))
    except Exception as e:
        print(f, 'failed', e)
for k, v in sorted(acc.items()): print(k, v)
PY
