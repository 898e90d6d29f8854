#!/bin/bash
# round 5, BEFORE the first gpurun call (runs here, no GPU, ~1 min): build the one candidate round 4 left measured-but-unshipped as a tagged
# variant from a scratch copy of the sources (product tree untouched): reset draws right after the integrator, Quadrotor3D only.
set -e
cd "$(dirname "$0")/../.." || exit 1
P=/tmp/scg_cand_q3after; S=$P/safe_control_gym_amd/csrc; rm -rf $P; mkdir -p $S; cp -r safe_control_gym_amd/csrc/. $S/; cp -r include $P/
patch -s -d $S scg_env_core.h < tools/candidates/r05_q3_reset_draws_after_integrator.patch
python tools/ab_variant.py build q3after --src $S --flags=-DSCG_EXP_PRE_AFTER --tasks quadrotor_3D_track
ls safe_control_gym_amd/spec/ | grep -c '_q3after.so'
