#!/bin/bash
# round 6 (second session) checkpoint: the whole GPU suite on this tree; learner A/Bs (forward layer-2 operand one block ahead in the one-tile gradient
# kernel; critic pass at 12 waves per workgroup = variant fw12); timeline of the one-tile form; the step-kernel rocprofv3 passes (kernel trace + PMC)
# and the learner profiles on this tree's sources
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s142; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_full.txt 2>&1; tail -6 $O/pytest_full.txt
run() { # label, env assignments...
  L=$1; shift
  env "$@" timeout 300 python tools/learner_profile.py ppo --iters 40 2>&1 | grep LEARNER_PROFILE | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('LEARNER_PROFILE ')[1]); print('$L', round(d['wall_ms_per_iteration'], 4), round(d['device_ms_per_iteration_median'], 4), d['last_update']['value_loss'])"
}
for rep in 1 2; do
  run "shipped (critic pass 8 waves)     " X=1
  run "critic pass 12 waves (fw12)       " SCG_LEARN_TAG=fw12
  run "previous library (round 5)        " SCG_LEARN_TAG=base
done 2>&1 | tee $O/ppo_ab.txt
SCG_LEARN_TAG=timing timeout 300 python tools/learn_cost.py --timeline --mb 16256 2>&1 | tail -18 | tee $O/timeline_one_tile.txt
timeout 600 python tools/sac_step_ab.py shipped base > $O/sac_ab.txt 2>&1; tail -4 $O/sac_ab.txt
SCG_PROFILE_SEQ=1 bash tools/profile_round5.sh > $O/profile5.log 2>&1; tail -3 $O/profile5.log
bash tools/profile_round6.sh > $O/profile6.log 2>&1; tail -4 $O/profile6.log | cut -c1-600
