#!/bin/bash
# round 3, session 1: the never-run pins (pybullet wheel probe; the reference's own PPO / SAC on HipVecEnv from the staged copy)
# + the PMC diagnosis of cartpole_stab's step kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s33; mkdir -p $O
{ date -u; echo '$ python -c "import pybullet"'; python -c "import pybullet, pybullet_data; print('pybullet', pybullet.getAPIVersion())" 2>&1
  echo '$ pip list | grep -i -E "bullet|gymnasium|casadi"'; pip list 2>/dev/null | grep -i -E "bullet|gymnasium|casadi" || echo "(none)"
  echo '$ find / -iname "*pybullet*"'; find / -xdev -iname "*pybullet*" -not -path "*/gpurun*" -not -path "$GRAFT_REPO_ROOT/*" 2>/dev/null | head; echo "(end)"; } > $O/pybullet_import.txt 2>&1
python tests/golden/pybullet_probe.py --write-trace > $O/pybullet_probe.log 2>&1
for A in ppo sac; do timeout 600 python tools/run_reference_ppo_on_hip.py --algo $A > $O/ref_$A.log 2>&1; echo "rc=$?" >> $O/ref_$A.log; done
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q > $O/pytest_dropin.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-secondary --ppo-seeds 0"
T=cartpole_stab; N=65536; i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/cp_diag$i -o p -- \
      python bench.py --task $T --envs $N --steps 100 --warmup 30 --no-graph $B > $O/cp_diag$i.log 2>&1 < /dev/null
  f=$(find $O/cp_diag$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $O/cp_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'step_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print(k, 'per launch', sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:])), 'n', len(v))
PY
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -size +2000k -delete
du -sh $O
