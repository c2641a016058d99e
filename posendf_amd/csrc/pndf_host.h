// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

// Every entry point runs on the device that owns its buffers and leaves the caller's current device as it found it
// (torch keeps a per-thread current device; an engine for cuda:1 must not change what torch.cuda.current_device() says).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (device < 0) return;                   // unknown: stay on the current device
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = (hipSetDevice(device) == hipSuccess);
        else prev = -1;                           // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// device that owns a device pointer (-1 if the runtime does not know it): the stateless helpers have no handle
inline int pndf_pointer_device(const void* p) {
    hipPointerAttribute_t attr;
    if (!p || hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return attr.device;
}
