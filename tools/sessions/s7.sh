#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/debug_update.py 1 2>&1 | grep -v amdgpu.ids | tail -12
python tools/debug_update.py 6 2>&1 | grep -v amdgpu.ids | tail -8
