#!/bin/bash
# round 4, BEFORE the first gpurun call (runs here, no GPU, ~3 min): build the round-3 candidates of tools/candidates/ as TAGGED variants of
# the specialised libraries (libscg_spec_<hash>_<tag>.so, git-ignored, travel with the snapshot) from scratch copies of the sources — the
# product tree stays untouched; SCG_SPEC_TAG=<tag> selects them on the box (tools/sessions/s81.sh).
set -e
cd "$(dirname "$0")/../.." || exit 1
R=$PWD
for v in "wide r04_wide_soa_row_stores.patch -DSCG_EXP_WIDE_ROWS" \
         "wide2 r04_wide_soa_row_stores.patch -DSCG_EXP_WIDE_ROWS=2" \
         "st17 r04_store_cache_policy.patch -DSCG_EXP_ST_AUX=17" \
         "recur r04_q2_recurrence_integrator.patch -DSCG_EXP_Q2_RECUR"; do
  set -- $v
  P=/tmp/scg_cand_$1; S=$P/safe_control_gym_amd/csrc; rm -rf $P; mkdir -p $S; cp -r safe_control_gym_amd/csrc/. $S/; cp -r include $P/   # (the sources include ../../include/*.h)
  patch -s -d $S scg_env_core.h < tools/candidates/$2
  python tools/ab_variant.py build $1 --src $S --flags=$3 --tasks quadrotor_2D_track,quadrotor_3D_track
done
# wide rows AND the recurrence integrator together (independent hunks)
P=/tmp/scg_cand_widerecur; S=$P/safe_control_gym_amd/csrc; rm -rf $P; mkdir -p $S; cp -r safe_control_gym_amd/csrc/. $S/; cp -r include $P/
patch -s -d $S scg_env_core.h < tools/candidates/r04_wide_soa_row_stores.patch
patch -s -d $S scg_env_core.h < tools/candidates/r04_q2_recurrence_integrator.patch && \
  python tools/ab_variant.py build widerecur --src $S "--flags=-DSCG_EXP_WIDE_ROWS -DSCG_EXP_Q2_RECUR" --tasks quadrotor_2D_track || echo "combined variant: patches do not compose cleanly, skipped"
ls safe_control_gym_amd/spec/ | grep -c '_wide.so\|_wide2.so\|_st17.so\|_recur.so\|_widerecur.so'
