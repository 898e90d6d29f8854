#!/bin/bash
# round 3, session 34: bench.py's timed region with the calibrated host wait (device synchronize vs event polling) and the
# per-rank clock that stops before the closing barrier — headline only, plain and as two gloo ranks on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s70; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/b1.json 2> $O/b1.err; python - <<'PY'
import json, os
d = json.load(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/s70/b1.json'))
print(d['value'], d['ms_per_step'], d['config']['timed_region_samples_ms'], d['config']['host_wait'], d['roofline']['avg_launch_us'])
PY
for m in device poll; do SCG_BENCH_HOST_WAIT=$m python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['config']['timed_region_samples_ms'])"; done
timeout 200 python -m pytest tests/test_gpu_multirank.py -x -q -k "bench_two_ranks or plain_python" 2>&1 | tail -2
