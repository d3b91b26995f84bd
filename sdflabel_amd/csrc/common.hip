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
extern "C" int sdfr_version(void) { return 200; }
