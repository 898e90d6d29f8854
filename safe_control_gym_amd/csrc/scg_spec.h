// scg_spec.h — included only by config-specialised builds (hipcc -DSCG_SPEC -include <generated>.h).
//
// The generated header defines SCG_SPEC_HASH / SCG_SPEC_SYS / SCG_SPEC_DTYPE / SCG_SPEC_DIST and
// SCG_SPEC_FILL(c): assignments of every CfgParams field as hexadecimal floating literals (exact).
// Here they become a `static constexpr CfgParams<T>` that device code reads as immediates.
#pragma once
#ifndef SCG_SPEC_FILL
#error "SCG_SPEC build needs the generated specialisation header (-include scg_spec_<hash>.h)"
#endif
#include "scg_params.h"

namespace scg {

template <typename T>
constexpr CfgParams<T> scg_make_spec_cfg() {
    CfgParams<T> c{};
    SCG_SPEC_FILL(c)
    return c;
}

template <typename T>
struct SpecHolder {
    static constexpr CfgParams<T> value = scg_make_spec_cfg<T>();
};

template <typename T>
__host__ __device__ constexpr const CfgParams<T>& scg_spec_cfg() {
    return SpecHolder<T>::value;
}

}  // namespace scg
