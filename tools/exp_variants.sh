#!/bin/bash
# Build timing-only variants of the specialised Q2-track f32 library (tools/exp_variants.sh build), then on the GPU box
# time each one through bench.py (tools/exp_variants.sh run).  Variants change results; they only attribute time.
H=$(python tools/timeline.py hash)
SPEC=safe_control_gym_amd/spec
case "$1" in
build)
  mkdir -p $SPEC/exp
  rm -f $SPEC/exp/*.so
  for v in BASE NO_RESET NO_CVAL NO_OBS ${EXTRA_VARIANTS}; do
    name=$(echo "$v" | sed 's/ -DSCG_EXP_/+/g; s/ -DSCG_/+/g; s/=/_/g')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DSCG_SPEC -DSCG_EXP_$v -include $SPEC/scg_spec_$H.h \
      -o "$SPEC/exp/$name.so" safe_control_gym_amd/csrc/scg_kernels.hip &
  done; wait; ls $SPEC/exp ;;
run)
  cp $SPEC/libscg_spec_$H.so /tmp/keep.so
  for f in $SPEC/exp/*.so; do
    cp "$f" $SPEC/libscg_spec_$H.so
    printf "%-60s " "$(basename $f .so)"
    timeout 120 python bench.py --no-cpu-baseline $2 < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f us/launch' % d['roofline']['avg_launch_us'])"
  done
  cp /tmp/keep.so $SPEC/libscg_spec_$H.so ;;
esac
