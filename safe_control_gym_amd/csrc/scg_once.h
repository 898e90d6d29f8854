// scg_once.h — one-time host-side work that must happen once PER HIP DEVICE (a kernel's MaxDynamicSharedMemorySize attribute
// is a per-device property): a process that drives envs / agents on several GPUs, or on a non-default one, sets it for each.
// Usage:   static scg::PerDeviceOnce once;  int d;  if (once.pending(&d)) { ...hipFuncSetAttribute...; once.commit(d); }
// Two threads racing on the same device both do the (idempotent) work; the flag is only ever set after the work succeeded.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace scg {
struct PerDeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool pending(int* dev) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        *dev = d & 63;
        return ((mask.load(std::memory_order_acquire) >> *dev) & 1ull) == 0;
    }
    void commit(int dev) { mask.fetch_or(1ull << dev, std::memory_order_release); }
};
}  // namespace scg
