// Shared host/device helpers for libsdfr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sdfr.h"

void sdfr_set_error(const char* fmt, ...);

// Compile-time options (kernel geometry overrides, the timing-only ablations SDFR_ABL_* / SDFR_PIN_WEIGHTS, the cycle trace) belong to
// experiment builds only: csrc/build.sh passes them with -DSDFR_EXPERIMENT (SDFR_AB=1) and sdfr_build_flags() reports it.
#if !defined(SDFR_EXPERIMENT) && (defined(SDFR_ABL_NOMFMA) || defined(SDFR_ABL_NOEPI) || defined(SDFR_PIN_WEIGHTS) || defined(SDFR_MLP_TRACE) || \
                                  defined(SDFR_EPI_FENCE) || defined(SDFR_MLP_WPE) || defined(SDFR_ACT_DBUF) || defined(SDFR_STRAIGHT_KLOOP) ||      \
                                  defined(SDFR_UNROLLED_KLOOP) || defined(SDFR_H_FAST_EPI) || defined(SDFR_BOX_PAD))
#error "experiment option without SDFR_EXPERIMENT: build through tools/ab_variant.sh (SDFR_AB=1)"
#endif

#define SDFR_HIP_CHECK(x)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            sdfr_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);     \
            (void)hipGetLastError();   /* reported through our return code: do not leave it as the   */ \
            return SDFR_E_HIP;         /* thread's sticky "last error" for the next HIP user (torch) */ \
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

// zero `bytes` (a multiple of 4) of device memory on `stream` with a KERNEL.  Not hipMemsetAsync: a memset node inside a captured HIP graph
// faulted on replay ("Memory access fault by GPU ... write access to a read-only page") as soon as any eager work -- a copy, an allocation --
// ran between two replays of the graph (ROCm 7.2 on gfx950, found in r04 with the traced refiner: profiles/r04_notes.md section 7); every
// entry point of this library may be captured, so none of them enqueues a memset.
hipError_t sdfr_zero_async(void* p, size_t bytes, hipStream_t stream);

// XCD-aware crop mapping of a (x = work item of a crop, y = crop) grid.  MI355X deals the workgroups of a launch round-robin to its 8 XCDs by
// their linear index, and each XCD has its own L2: with the plain mapping the workgroups of ONE crop land on all eight and every L2 fetches
// that crop's data (surfel arrays in the splat forward, pixel records in its backward) for itself.  Re-deal the indices so that crop 8g + x
// is served by XCD x alone.  Crops beyond the last full group of 8 keep the plain mapping.  A bijection of the grid: kernels whose results
// do not depend on which workgroup computes what (all of ours) return the same bits.
#define SDFR_XCDS 8
#ifdef __HIPCC__
__device__ __forceinline__ void sdfr_xcd_crop_map(int& xb, int& b) {
    const int nbx = gridDim.x, full = (gridDim.y / SDFR_XCDS) * SDFR_XCDS;
    const int64_t lin = (int64_t)blockIdx.y * nbx + blockIdx.x;
    xb = blockIdx.x; b = blockIdx.y;
    if (lin < (int64_t)nbx * full) {
        const int64_t j = lin / SDFR_XCDS;
        b = (int)(j / nbx) * SDFR_XCDS + (int)(lin % SDFR_XCDS);
        xb = (int)(j % nbx);
    }
}
#endif

// number of valid items of crop b in a [B][cap] ragged array
__device__ __forceinline__ int sdfr_count(const int32_t* cnt, int b, int cap) {
    if (cnt == nullptr) return cap;
    int c = cnt[b];
    return c < cap ? c : cap;
}
