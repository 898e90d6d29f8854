#!/bin/bash
# round 5, final: the whole GPU suite at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s124
( time timeout 560 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/s124/pytest_full.txt 2>&1; tail -6 gpurun_out/s124/pytest_full.txt
