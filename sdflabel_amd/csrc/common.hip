#include "sdfr_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void sdfr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(256) void sdfr_zero_kernel(uint32_t* __restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) p[i] = 0u;
}

hipError_t sdfr_zero_async(void* p, size_t bytes, hipStream_t stream) {
    const size_t words = bytes / 4;
    if (words == 0) return hipSuccess;
    const size_t blocks = (words + 255) / 256;
    hipLaunchKernelGGL(sdfr_zero_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, (uint32_t*)p, words);
    return hipGetLastError();
}

extern "C" const char* sdfr_last_error(void) { return g_err; }
// 300: r05 -- sdfr_trace_march / sdfr_trace_cone took their r04 argument lists (levels array, q_max, spec_k, sigma, aux lists) and
// SDFR_TRACE_COUNTERS grew from 8 to 32 words under version 200; a caller built against that header must not bind this library silently
// 400: r06 -- fused entry points (sdfr_params_plan, sdfr_band_select_ex, sdfr_mlp_forward_candidates, sdfr_candidate_band, sdfr_losses_fused,
// sdfr_splat_backward_x, sdfr_pose_latent_solver)
extern "C" int sdfr_version(void) { return SDFR_VERSION; }

// bit 0: experiment build (SDFR_EXPERIMENT: some kernel geometry or option differs from the product's); bit 1: a timing-only ablation is
// compiled in (results are WRONG by construction).  The product library returns 0; sdflabel_amd/_lib.py refuses anything else.
extern "C" int sdfr_build_flags(void) {
    int f = 0;
#ifdef SDFR_EXPERIMENT
    f |= 1;
#if defined(SDFR_ABL_NOMFMA) || defined(SDFR_ABL_NOEPI) || defined(SDFR_PIN_WEIGHTS)
    f |= 2;
#endif
#endif
    return f;
}
