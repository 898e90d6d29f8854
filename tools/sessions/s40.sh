#!/bin/bash
# round 3, session 8: same-box A/B of step-kernel variants (prebuilt in-tree, tagged): base (PreDraw on, default scheduler), ilp
# (-amdgpu-sched-strategy=max-ilp), nopre / ilpnopre (Q3), iterilp, memc
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s40; mkdir -p $O
B="--no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0 --steps 20000 --warmup 2000"
for rep in 1 2; do
for T in quadrotor_2D_track cartpole_stab quadrotor_3D_track quadrotor_3D_track_disturbed; do
  for V in base ilp nopre ilpnopre iterilp memc; do
    ls safe_control_gym_amd/spec/ | grep -q "_$V.so" || continue
    SCG_SPEC_TAG=$V timeout 120 python bench.py --task $T $B > $O/${T}__${V}__$rep.json 2>> $O/err.log
  done
done
done
python - <<'PY'
import json, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/s40/*.json'))):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][0])
        key = os.path.basename(f).rsplit('__', 1)[0]
        acc[key].append((round(d['roofline']['avg_launch_us'], 3), d['config']['kernel_build'][:7]))
    except Exception as e:
        pass
for k, v in sorted(acc.items()): print(k, v)
PY
