#!/bin/bash
# round 5, GPU session 5: the rocprofv3 passes behind profiles/r05_* (step kernels incl. the 4 M-env streaming point, learner iterations).
cd "$GRAFT_REPO_ROOT" || exit 1
SCG_PROFILE_LEARNERS=1 bash tools/profile_round5.sh 2>&1 | tail -40
