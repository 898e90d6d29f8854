#!/bin/bash
# rocprofv3 passes behind profiles/r01_*: run ON THE GPU BOX (gpurun -- 'tools/profile_round.sh'), writes gpurun_out/prof/.
# Counter passes are separate from the kernel trace (and from each other: TCC slot limit), as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof; mkdir -p $OUT
for N in 65536 1048576; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$N -o p -- \
      python bench.py --envs $N --steps 2000 --warmup 200 --no-cpu-baseline > $OUT/kt_$N.log 2>&1 < /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$N -o p -- \
        python bench.py --envs $N --steps 200 --warmup 50 --no-cpu-baseline --no-graph > $OUT/pmc_${C}_$N.log 2>&1 < /dev/null
  done
done
ls -R $OUT | head -40
