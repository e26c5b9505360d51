// Dynamic LDS beyond the 64 KiB default must be granted to a kernel with hipFuncSetAttribute.  The grant is per device and only
// ever needs raising: this helper remembers, per kernel (template argument) and device ordinal, the largest size granted so
// far and calls the runtime only to raise it.  (Rounds 1 - 2 cached a per-process flag: wrong for a second device, racy between
// host threads; round 3 first set the attribute on every launch: 20 runtime calls per small transform, which is launch-bound.)
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace atlas_amd {

template <auto Kernel>
inline hipError_t ensure_dynamic_lds(int bytes) {
    constexpr int MAX_DEV = 64;
    static std::atomic<int> granted[MAX_DEV];   // zero-initialised
    int dev      = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        return e;
    }
    const bool tracked = dev >= 0 && dev < MAX_DEV;
    if (tracked && granted[dev].load(std::memory_order_acquire) >= bytes) {
        return hipSuccess;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && tracked) {
        int old = granted[dev].load(std::memory_order_relaxed);
        while (old < bytes && !granted[dev].compare_exchange_weak(old, bytes, std::memory_order_release)) {
        }
    }
    return e;
}

}  // namespace atlas_amd
