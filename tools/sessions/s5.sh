#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s5; mkdir -p $O
for mode in "" "--no-fused"; do
  timeout 100 python examples/train_ppo.py --max-seconds 6 --seed 3 $mode 2>/dev/null > $O/run$mode.jsonl
  python - "$O/run$mode.jsonl" <<'PY'
import sys, json
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{') and 'summary' not in l]
print(sys.argv[1], len(rows))
for r in rows[:6]+rows[20:24]+rows[50:54]:
    print({k: r.get(k) for k in ('policy_loss','value_loss','entropy_loss','approx_kl','actor_steps','eval_return','ep_return')})
PY
done
