// kr_lds_optin.h -- more than 64 KiB of dynamic LDS per workgroup (gfx950: 160 KiB per CU) is an opt-in, and
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives several GPUs (one engine per device,
// or one process per GPU that touches a second ordinal) must set it on every device it launches on.  One table for the whole library, keyed by
// (device, kernel); a request only ever raises the window.  Not a stream operation: call sites inside a captured decode step go through their
// *_prepare functions, which run before the capture.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

inline int kr_lds_optin(const void* fn, size_t bytes) {      // 0 = the window of `fn` on the current device is >= bytes
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> window;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(dev, fn);
    const auto it = window.find(key);
    if (it != window.end() && it->second >= bytes) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 1;
    window[key] = bytes;
    return 0;
}
