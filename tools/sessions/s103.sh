#!/bin/bash
# round 5, first GPU session (≈ 9 GPU-minutes; run tools/sessions/s103_prepare.sh HERE first):
#   1. the whole GPU suite + smoke + the driver-style bench on the tree round 4 ended with (last seen green: tools/sessions/s99.sh / s100.sh);
#   2. the Quadrotor3D reset-draw candidate, same box, alternating (round 4: 8.05 vs 8.20 us, -1.8 %): ship it (product macro, full parity suite,
#      SCG_PROFILE_ENV_ONLY=1 bash tools/profile_round4.sh + python tools/profile_post.py <tag>) or reject it with these numbers.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s103; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -rxXs ) > $O/suite.log 2>&1; tail -6 $O/suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 200 python tools/ab_variant.py run q3after --tasks quadrotor_3D_track --rounds 3 2>&1 | tee $O/ab_q3after.log | grep 'tag=\|one-step' | cut -c1-250
