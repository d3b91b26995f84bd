// Shared host/device helpers for libsdfr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sdfr.h"

void sdfr_set_error(const char* fmt, ...);

#define SDFR_HIP_CHECK(x)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            sdfr_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);     \
            return SDFR_E_HIP;                                                                         \
        }                                                                                              \
    } while (0)

#define SDFR_REQUIRE(cond, ...)                                                                        \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            sdfr_set_error(__VA_ARGS__);                                                               \
            return SDFR_E_INVALID;                                                                     \
        }                                                                                              \
    } while (0)

#define SDFR_LAUNCH_CHECK() SDFR_HIP_CHECK(hipGetLastError())

// makes `device` current for the lifetime of the guard and restores the caller's device afterwards (a library call must not change the
// process' current device: PyTorch and other HIP users rely on it)
struct SdfrDeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit SdfrDeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device); else if (err == hipSuccess) prev = -1;
    }
    ~SdfrDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static inline int sdfr_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// number of valid items of crop b in a [B][cap] ragged array
__device__ __forceinline__ int sdfr_count(const int32_t* cnt, int b, int cap) {
    if (cnt == nullptr) return cap;
    int c = cnt[b];
    return c < cap ? c : cap;
}
