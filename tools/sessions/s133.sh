#!/bin/bash
# round 6 (second session): SAC step with stored activations (gradient kernels start at the loss derivatives), the online critics' forward next to
# the targets, step k + 1's first launch inside step k's target-action launch (scg_sac_update_n), reductions with every load in flight
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s133; mkdir -p $O
timeout 600 python tools/sac_step_ab.py shipped base > $O/sac_ab.txt 2>&1; tail -8 $O/sac_ab.txt
( time timeout 1200 python -m pytest tests/test_gpu_sac_fused.py tests/test_gpu_rl.py tests/test_gpu_multirank.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
