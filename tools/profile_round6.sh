#!/bin/bash
# rocprofv3 passes behind profiles/r06_*: run ON THE GPU BOX (gpurun -- 'bash tools/profile_round6.sh'), writes gpurun_out/prof6/.
#   learner iterations (tools/learner_profile.py): the PPO leg's iteration (one HIP-graph replay) and the SAC leg's vector step — kernel
#   trace + stats, then the same commands WITHOUT the tracer for the wall clock; condensed on the box by tools/learner_profile_post.py
#   into r06_learner_kernel_sums.json (what bench.py's ppo.iteration_ms.kernel_sum / sac.roofline quote, keyed by the source hashes).
#   SCG_PROFILE_STEP=1: also the round-5 step-kernel passes (tools/profile_round5.sh's specs) into the same directory.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof6; rm -rf $OUT; mkdir -p $OUT
for M in ppo sac; do
  IT=40; [ $M = sac ] && IT=200
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$M -o p -- \
      python tools/learner_profile.py $M --iters $IT > $OUT/kt_$M.log 2>&1 < /dev/null
  timeout 300 python tools/learner_profile.py $M --iters $IT > $OUT/plain_$M.log 2>&1 < /dev/null
done
timeout 300 python tools/learner_profile.py ppo --iters 40 --no-graph > $OUT/plain_ppo_nograph.log 2>&1 < /dev/null
python tools/learner_profile_post.py $OUT
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*.db' -delete; find $OUT -name '*.log' -size +200k -delete
du -sh $OUT
