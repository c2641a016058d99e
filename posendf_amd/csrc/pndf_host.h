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

// roctx ranges around the C-ABI compute entry points: `rocprofv3 --marker-trace` then shows
// pndf_forward / pndf_forward_grad / pndf_project / pndf_lbs_terms_grad ... as named host ranges next to the kernels they
// launch.  The library does NOT link against the profiler: the marker library is picked up when the process already has it
// loaded (rocprofv3 preloads it) or when PNDF_ROCTX=1 asks for it; otherwise a range is two null checks.
#include <dlfcn.h>
#include <stdlib.h>
struct PndfRoctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    PndfRoctx() {
        const char* env = getenv("PNDF_ROCTX");
        if (env && env[0] == '0') return;
        const bool force = env && env[0] == '1';
        const char* libs[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        for (const char* name : libs) {
            void* lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (!lib && force) lib = dlopen(name, RTLD_NOW);
            if (!lib) continue;
            push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
            pop = (int (*)())dlsym(lib, "roctxRangePop");
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
inline PndfRoctx& pndf_roctx() {
    static PndfRoctx r;
    return r;
}
struct PndfRange {
    bool on;
    explicit PndfRange(const char* name) : on(pndf_roctx().push != nullptr) {
        if (on) pndf_roctx().push(name);
    }
    ~PndfRange() {
        if (on) pndf_roctx().pop();
    }
    PndfRange(const PndfRange&) = delete;
    PndfRange& operator=(const PndfRange&) = delete;
};
