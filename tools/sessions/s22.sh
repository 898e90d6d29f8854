#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for fl in "-DDBG_DW1_PLAIN" "-DDBG_NO_DW2 -DDBG_NO_DW3 -DDBG_DW1_PLAIN" "-DDBG_NO_DW2 -DDBG_NO_DW3"; do
  SCG_LEARN_FLAGS="$fl" python -c "
from safe_control_gym_amd import _learn
_learn.build(12,128,2,'tanh', force=True)" > /dev/null 2>&1
  echo "== flags: $fl"; python tools/learn_cost.py 2>&1 | grep "per tile\|65536"
done
