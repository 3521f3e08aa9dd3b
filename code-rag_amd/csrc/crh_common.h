// crh_common.h -- shared host-side helpers for libcoderag_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/coderag_hip.h"

namespace crh {

std::string &last_error_ref();

inline int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define CRH_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::crh::fail(CRH_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                            \
    } while (0)

#define CRH_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != CRH_OK) return rc_; \
    } while (0)

// RAII guard: make `device` current for the duration of a call, restore on exit.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute (dynamic LDS above 64 KiB) applies to the CURRENT device only: one flag per device, not per process.
// need() is true the first time it is asked on a device.  (A race between two threads repeats an idempotent call.)
struct OncePerDevice {
    bool done[64] = {};
    bool need()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return true;
        d &= 63;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// CU count of the CURRENT device, looked up once per device (a process-wide static would hand device 0's count to a
// launch on another device of a mixed node).
inline int current_device_cus()
{
    static int cus[64] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return 256;
    d &= 63;
    if (!cus[d]) {
        hipDeviceProp_t prop;
        cus[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount >= 8) ? prop.multiProcessorCount : 256;
    }
    return cus[d];
}

}  // namespace crh
