#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/debug_learn.py 524288 65536 2>&1 | tail -20
python tools/debug_learn.py 8192 4096 2>&1 | tail -20
