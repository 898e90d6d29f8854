#!/bin/bash
# (historical: SCG_PAIR_MAX_ENVS and the pe64 / pe128 / spec tags belong to the paired launch, which this session measured and the round then removed)
# round 5, GPU session 2: the paired step launch (bitwise tests, A/B at 65 536 envs for the four tasks, workgroup-size / speculation variants, N sweep),
# workgroup sizes in the streaming regime (1 M - 16 M envs), the graph-captured data-parallel epoch over one-rank RCCL, timelines.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s107; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_split_step.py tests/test_gpu_kstep_oracle.py "tests/test_gpu_multirank.py::test_graph_captured_data_parallel_epoch_over_rccl_with_one_rank" -x -q 2>&1 | tail -15 | tee $O/pytest_a.txt
B="--steps 4000 --warmup 500 --no-secondary --no-cpu-baseline --ppo-seeds 0 --sac-seeds 0"
one() {  # task envs label [env assignments...]
  local task=$1 envs=$2 label=$3; shift 3
  env "$@" python bench.py --task $task --envs $envs $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-30s %9d %-12s %.4f us  frac %.4f  finite=%s' % ('$task', $envs, '$label', r['avg_launch_us'], r['frac'] or 0, d['config']['finite_outputs']))"
}
PLAIN="SCG_SPLIT_MAX_ENVS=0 SCG_PAIR_MAX_ENVS=0 SCG_WIDE_MIN_ENVS=2000000000"
for rep in 1 2; do
for t in quadrotor_2D_track cartpole_stab quadrotor_3D_track quadrotor_3D_track_disturbed; do
  one $t 65536 pair SCG_X=1; one $t 65536 plain $PLAIN
done; done 2>&1 | tee $O/ab_pair.txt
for tag in pe64 pe128 spec spec64; do one quadrotor_2D_track 65536 "tag=$tag" SCG_SPEC_TAG=$tag; done 2>&1 | tee $O/ab_pair_variants.txt
for t in cartpole_stab quadrotor_3D_track; do one $t 65536 "tag=spec" SCG_SPEC_TAG=spec; done 2>&1 | tee -a $O/ab_pair_variants.txt
for n in 16384 32768 49152 98304 131072; do
  one quadrotor_2D_track $n pair SCG_SPLIT_MAX_ENVS=0 SCG_PAIR_MAX_ENVS=2000000000; one quadrotor_2D_track $n split SCG_SPLIT_MAX_ENVS=2000000000; one quadrotor_2D_track $n plain $PLAIN
done 2>&1 | tee $O/ab_pair_n.txt
for n in 1048576 2097152 4194304 16777216; do
  for tag in "" b128 b256 b512 b1024 wb wb256; do
    one quadrotor_2D_track $n "tag=$tag" SCG_SPEC_TAG=$tag $PLAIN
  done
done 2>&1 | tee $O/ab_stream.txt
for tag in "" b256 b512; do one quadrotor_2D_track 4194304 "tag=$tag r2" SCG_SPEC_TAG=$tag $PLAIN; done 2>&1 | tee -a $O/ab_stream.txt
for t in cartpole_stab quadrotor_3D_track; do for tag in "" b256; do one $t 4194304 "tag=$tag" SCG_SPEC_TAG=$tag $PLAIN; done; done 2>&1 | tee -a $O/ab_stream.txt
timeout 120 python tools/timeline.py run 65536 pair > $O/timeline_pair.txt 2>&1; grep -v amdgpu.ids $O/timeline_pair.txt
timeout 120 python tools/timeline.py run 65536 plain > $O/timeline_plain.txt 2>&1; grep -A10 "every output" $O/timeline_plain.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --ppo-seeds 0 --sac-seeds 0 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s107/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['frac_by_clock'])
print('secondary', {k: (v.get('avg_launch_us'), v.get('frac')) for k, v in d.get('secondary', {}).items()})
print('f64', d.get('f64', {}).get('avg_launch_us'), d.get('f64', {}).get('frac'))
print('sequence', {k: (v.get('us_per_control_step'), v.get('frac')) for k, v in d.get('sequence', {}).items() if isinstance(v, dict)})
print('multi_gpu', d.get('multi_gpu'))
PY
