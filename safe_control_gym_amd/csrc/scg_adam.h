// scg_adam.h — torch.optim.Adam's element update (betas 0.9 / 0.999, eps 1e-8, no weight decay: the optimisers of controllers/ppo/ppo_utils.py:50-56 and
// controllers/sac/sac_utils.py:52-58) and the soft target update (sac_utils.py:166-168), with every product rounded where it is WRITTEN.
// Left to the compiler, which multiply of  b m + (1 - b) g  is fused into the fma depends on instruction scheduling (an operand that
// arrives early gets its product issued early and rounded on its own): kernels that must agree bit for bit — the fused reduction + step
// against gradient reduction then step, the data-parallel step against the single-GPU one — then differ in the last place
// (tests/test_gpu_learn.py caught exactly that when the moments' loads moved to the top of the fused kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace scg {

__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, float lr, float t) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    m = __builtin_fmaf(1.0f - b1, g, __fmul_rn(b1, m));
    v = __builtin_fmaf(__fmul_rn(1.0f - b2, g), g, __fmul_rn(b2, v));
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    p = p - __fmul_rn(lr / bc1, m) / (sqrtf(v) / sqrtf(bc2) + eps);
}

// target <- (1 - tau) target + tau p
__device__ __forceinline__ float polyak(float target, float p, float tau) { return __builtin_fmaf(tau, p, __fmul_rn(1.0f - tau, target)); }

}  // namespace scg
