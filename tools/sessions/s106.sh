#!/bin/bash
# round 5, GPU session 1: issue-rate microbenchmark; bitwise tests of the split step launch + the K-step kernels against the oracle; A/B of the
# split launch (four tasks, threshold probe); streaming-regime variants at 1 M / 4 M envs; timeline of the new kernel; driver-style headline.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s106; mkdir -p $O
python -c "import pybullet" 2>&1 | tail -1 > $O/pybullet_probe.txt
rocm-smi --showclocks --showpower > $O/rocm_smi.txt 2>&1
tools/issue_rate > $O/issue_rate.txt 2>&1; head -40 $O/issue_rate.txt
timeout 600 python -m pytest tests/test_gpu_split_step.py tests/test_gpu_kstep_oracle.py tests/test_gpu_sequence.py -x -q 2>&1 | tail -15 | tee $O/pytest_a.txt
timeout 500 python -m pytest tests/test_gpu_env_parity.py -x -q 2>&1 | tail -8 | tee $O/pytest_b.txt
B="--steps 4000 --warmup 500 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
one() {  # task envs label [env assignments...]
  local task=$1 envs=$2 label=$3; shift 3
  env "$@" python bench.py --task $task --envs $envs $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-30s %9d %-10s %.4f us  frac %.4f  finite=%s' % ('$task', $envs, '$label', r['avg_launch_us'], r['frac'] or 0, d['config']['finite_outputs']))"
}
for rep in 1 2; do
for t in quadrotor_2D_track cartpole_stab quadrotor_3D_track quadrotor_3D_track_disturbed; do
  one $t 65536 split SCG_X=1; one $t 65536 nosplit SCG_SPLIT_MAX_ENVS=0
done; done 2>&1 | tee $O/ab_split.txt
for n in 32768 98304 131072 196608 262144; do
  one quadrotor_2D_track $n split SCG_SPLIT_MAX_ENVS=1000000000; one quadrotor_2D_track $n nosplit SCG_SPLIT_MAX_ENVS=0
done 2>&1 | tee $O/ab_split_n.txt
for n in 1048576 4194304; do
  for tag in "" wb b256 wb256 dsched wbd; do
    one quadrotor_2D_track $n "tag=$tag" SCG_SPEC_TAG=$tag SCG_SPLIT_MAX_ENVS=0
  done
done 2>&1 | tee $O/ab_stream.txt
timeout 120 python tools/timeline.py run 65536 > $O/timeline_split.txt 2>&1; cat $O/timeline_split.txt | grep -v amdgpu.ids
timeout 120 python tools/timeline.py run 65536 nosplit > $O/timeline_nosplit.txt 2>&1; grep -A12 "every output" $O/timeline_nosplit.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --ppo-seeds 0 --sac-seeds 0 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s106/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['frac_by_clock'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('frac')) for k, v in d.get('secondary', {}).items()})
print('f64', d.get('f64', {}).get('avg_launch_us'), d.get('f64', {}).get('frac'))
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
PY
