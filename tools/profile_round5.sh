#!/bin/bash
# rocprofv3 passes behind profiles/r05_*: run ON THE GPU BOX (gpurun -- 'bash tools/profile_round5.sh'), writes gpurun_out/prof/
# (condensed HERE by `python tools/profile_post.py r05`).  Counter passes are separate from the kernel trace and from each other.
#   step kernels (one launch per control step; <= 98 304 envs: the split launch, step_split_kernel): the four shipped tasks at
#       65 536 envs f32 + the headline in float64 + the headline at 4 194 304 envs (the streaming regime):
#       --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_INSTS_VALU SQ_WAVES; at 4 M envs also the wave-cycle / wait /
#       L2 request counters (what the launch waits for)
#   K-steps-per-launch kernels (tools/seq_profile.py): as in round 4     (SCG_PROFILE_SEQ=1 to include them)
#   learner iterations (PPO 65 536 envs, SAC): kernel trace only         (SCG_PROFILE_LEARNERS=1 to include them)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-secondary --ppo-seeds 0 --sac-seeds 0"
for spec in ${SCG_PROFILE_SPECS:-quadrotor_2D_track:65536:f32 cartpole_stab:65536:f32 quadrotor_3D_track:65536:f32 quadrotor_3D_track_disturbed:65536:f32 quadrotor_2D_track:65536:f64 quadrotor_2D_track:4194304:f32}; do
  IFS=: read T N DT <<< "$spec"
  STEPS=2000; [ "$N" -gt 1000000 ] && STEPS=200
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${T}_${DT}_$N -o p -- \
      python bench.py --task $T --envs $N --dtype $DT --steps $STEPS --warmup 200 $B > $OUT/kt_${T}_${DT}_$N.log 2>&1 < /dev/null
  CS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES")
  [ "$N" -gt 1000000 ] && CS+=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
  for C in "${CS[@]}"; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C// /+}_${T}_${DT}_$N" -o p -- \
        python bench.py --task $T --envs $N --dtype $DT --steps 100 --warmup 30 --no-graph $B > "$OUT/pmc_${C// /+}_${T}_${DT}_$N.log" 2>&1 < /dev/null
  done
done
if [ -n "$SCG_PROFILE_SEQ" ]; then
for M in sequence_all sequence_collector rollout_policy; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${M}_f32_65536 -o p -- \
      python tools/seq_profile.py --mode $M --reps 60 > $OUT/kt_$M.log 2>&1 < /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_${M}_f32_65536 -o p -- \
        python tools/seq_profile.py --mode $M --reps 30 > $OUT/pmc_${C}_$M.log 2>&1 < /dev/null
  done
done
fi
if [ -n "$SCG_PROFILE_LEARNERS" ]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_iteration_65536 -o p -- \
    python tools/ppo_profile.py --fused-rollout --envs 65536 --iters 20 --epochs 2 --minibatch 16256 --mb-per-epoch 32 > $OUT/ppo_iteration_65536.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sac_iteration -o p -- \
    python tools/sac_time_to_reward.py --budget 12 --eval-every 100000 > $OUT/sac_iteration.log 2>&1 < /dev/null
fi
# counter CSVs carry one row per dispatch (and counter dimension) of EVERY kernel: several MB per pass.  They are condensed ON THE BOX to
# one row per (kernel, counter) — the mean over the second half of the dispatches of the per-dispatch sum — in p_counter_collection.csv's own
# column names (what tools/profile_post.py reads), so that gpurun can merge the directory back (<= 64 MiB).
python - <<'PY'
import collections, csv, glob, os
for f in glob.glob(os.path.join('gpurun_out', 'prof', '**', '*counter_collection.csv'), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))        # (kernel, counter) -> dispatch -> sum over rows
    meta = {}
    for r in csv.DictReader(open(f)):
        k = (r.get('Kernel_Name', ''), r.get('Counter_Name', ''))
        if 'scg::' not in k[0] and 'step_' not in k[0]:
            continue
        per[k][r.get('Dispatch_Id', '0')] += float(r.get('Counter_Value', 0) or 0)
        meta[k] = (r.get('VGPR_Count', ''), r.get('SGPR_Count', ''), r.get('Grid_Size', ''), r.get('Workgroup_Size', ''))
    with open(f, 'w', newline='') as g:
        w = csv.writer(g)
        w.writerow(['Kernel_Name', 'Counter_Name', 'Counter_Value', 'Dispatches_Averaged', 'Dispatches_Total', 'VGPR_Count', 'SGPR_Count', 'Grid_Size', 'Workgroup_Size'])
        for k, d in per.items():
            ids = sorted(d, key=lambda x: int(x) if x.isdigit() else 0)
            half = ids[len(ids) // 2:]
            w.writerow([k[0], k[1], sum(d[i] for i in half) / max(1, len(half)), len(half), len(ids)] + list(meta[k]))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*.db' -delete; find $OUT -name '*.log' -size +200k -delete
du -sh $OUT; du -sk $OUT/* | sort -n | tail -5
