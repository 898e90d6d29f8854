#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for a in "fused eval" "fused noeval" "torch eval"; do python tools/debug_flow.py $a 2>&1 | grep -v amdgpu.ids | tail -9; done
