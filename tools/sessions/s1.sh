#!/bin/bash
# GPU session 1 (round 2): parity of the restructured kernels, first timings, sub-shard launch experiment, Q3 unroll sweep.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s1; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for t in quadrotor_2D_track quadrotor_3D_track quadrotor_3D_track_disturbed cartpole_stab; do
  timeout 200 python bench.py --task $t --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_$t.json 2> $O/bench_$t.err
  python -c "import json,sys; d=json.loads(open('$O/bench_$t.json').read().strip().split('\n')[-1]); print('$t', round(d['roofline']['avg_launch_us'],3),'us', d['config']['kernel_build'], d['config']['finite_outputs'])"
done
for u in 1 2 5 10 20; do
  SCG_SPEC_TAG=u$u timeout 200 python bench.py --task quadrotor_3D_track --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_q3_u$u.json 2> $O/bench_q3_u$u.err
  python -c "import json,sys; d=json.loads(open('$O/bench_q3_u$u.json').read().strip().split('\n')[-1]); print('q3 unroll $u', round(d['roofline']['avg_launch_us'],3),'us', d['config']['kernel_build'])"
done
for t in quadrotor_2D_track quadrotor_3D_track; do
  timeout 300 python tools/shard_bench.py --task $t --steps 4000 > $O/shard_$t.jsonl 2> $O/shard_$t.err
  cat $O/shard_$t.jsonl
done
