#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for mode in "" "--no-fused" "" "--no-fused" "--no-graphs"; do
  timeout 100 python examples/train_ppo.py --max-seconds 1.0 --seed 3 $mode 2>/dev/null | head -4 | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'summary' in r: continue
    print('[$mode]', {k: r.get(k) for k in ('policy_loss','value_loss','approx_kl','actor_steps','eval_return')})
"
done
