#!/bin/bash
# round 5, GPU session 9: same-box A/B of the learner micro-changes (tag old = the previous commit's scg_learn.hip + scg_mlp.h)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s114; mkdir -p $O
for rep in 1 2; do for tag in "" old; do echo "== SCG_LEARN_TAG=[$tag]"; SCG_LEARN_TAG=$tag timeout 200 python tools/learn_cost.py 2>&1 | grep -v amdgpu.ids; done; done | tee $O/learn_cost_ab.txt
