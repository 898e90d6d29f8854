#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s20; mkdir -p $O
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from tests.test_gpu_learn import _agent, _data
ag = _agent(12, 128, 2, 'tanh'); M = 524288; data = _data(12, 2, M, ag)
F = ag._build_fused(data, 65536); F['idx'].copy_(torch.randperm(M, device='cuda')[:65536].to(torch.int32))
for _ in range(40):
    ag._fused_grad(F); ag._fused_adam(F)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o learn -- python /tmp/one.py > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
