#!/bin/bash
# round 3, session 36: per-env deltas of the fuzzed 1-D quadrotor configs whose dynamics channel moves the state by ~1e-9
cd "$GRAFT_REPO_ROOT" || exit 1
for a in "quadrotor_1D 7 2" "quadrotor_1D 7 2 disturbances=None" "quadrotor_1D 1 2" "quadrotor_1D 3 2" "quadrotor_1D 3 2 disturbances={'dynamics':[{'disturbance_func':'step','magnitude':0.07,'step_offset':0}]}" "quadrotor_3D 5 30"; do
  echo "== $a"; python tools/fuzz_diag.py $a 2>&1 | tail -5 | cut -c1-400
done
