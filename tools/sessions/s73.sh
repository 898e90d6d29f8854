#!/bin/bash
# round 3, session 37: single-env facade vs the oracle's un-vectorised step on random configs (+ the 3-D free-running case of s71)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s73; mkdir -p $O
timeout 250 python -m pytest tests/test_gpu_config_fuzz.py -q -k "facade or (f64_generic and quadrotor_3D)" 2>&1 | tee $O/facade.log | grep -E "passed|failed|^E  +(AssertionError|assert|Mismatch|Max abs| ACTUAL| DESIRED|.*seed=)" | cut -c1-300 | head -60
