#!/bin/bash
# round 5: the two tests added after the full-suite run (SAC data-parallel steps captured over one-rank RCCL; Safe-Explorer pre-training resume) + the multirank file
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s117; mkdir -p $O
timeout 500 python -m pytest "tests/test_gpu_multirank.py::test_graph_captured_data_parallel_sac_steps_over_rccl_with_one_rank" "tests/test_gpu_multirank.py::test_graph_captured_data_parallel_epoch_over_rccl_with_one_rank" tests/test_gpu_dropin.py -x -q 2>&1 | tail -15 | tee $O/pytest.txt
