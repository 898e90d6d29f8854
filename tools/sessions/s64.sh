#!/bin/bash
# round 3, session 28: PMC passes on the rewritten ppo_grad_kernel (same counter sets as round 2, tools/sessions/s32.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s64; mkdir -p $O
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from tests.test_gpu_learn import _agent, _data
ag = _agent(12, 128, 2, 'tanh'); M = 524288; data = _data(12, 2, M, ag)
F = ag._build_fused(data, 65024); F['idx'].copy_(torch.randperm(M, device='cuda')[:65024].to(torch.int32))
for _ in range(12):
    ag._fused_grad(F); ag._fused_adam(F)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$O/$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $d -o p -- python /tmp/one.py > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ppo_grad' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print(k, 'per launch', sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:])), 'n', len(v))
PY
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_learn.py tests/test_gpu_sac_fused.py tests/test_gpu_rl.py tests/test_gpu_rollout_policy.py -x -q 2>&1 | tail -3
