// Micro-benchmark: every workgroup streams the SAME buffer (like the decoder kernels stream the weight image) out of L2 / Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_stream tools/micro/l2_stream.hip && /tmp/l2_stream
// Variants: lockstep (all workgroups walk the buffer from 0), staggered (workgroup g starts at chunk (g * stride) mod nchunks),
// waves of a workgroup on 8 interleaved slices or on one contiguous stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int MODE>   // 0 lockstep, 1 staggered by workgroup, 2 staggered by XCD-local CU index
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ buf, int n16, int passes, int stagger, uint4* __restrict__ out) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // wave w reads slice w (n16/8 uint4 each), 64 lanes x 16 B = 1 KiB per load instruction, UNROLL loads in flight
    const int per = n16 / 8;
    const uint4* base = buf + (size_t)wave * per;
    const int steps = per / 64;
    int start = 0;
    if (MODE == 1) start = (int)(((long long)blockIdx.x * stagger) % steps);
    if (MODE == 2) start = (int)(((long long)(blockIdx.x >> 3) * stagger) % steps);     // consecutive workgroup ids go round-robin over the 8 XCDs
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int p = 0; p < passes; ++p) {
        int s = start;
        for (int i = 0; i < steps; i += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int idx = s + u; if (idx >= steps) idx -= steps;
                v[u] = base[(size_t)idx * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
            s += 8; if (s >= steps) s -= steps;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 7u) out[blockIdx.x * 512 + tid] = acc;     // never true for the fill pattern; keeps the loads alive
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 512;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("%s, %d CUs, clock %d MHz, L2 %d KiB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.l2CacheSize / 1024);
    uint4* out; hipMalloc(&out, (size_t)wgs * 512 * 16);
    const double sizes_mb[] = {0.5, 1, 2, 3.5, 4, 7, 14, 28};
    for (double mb : sizes_mb) {
        int n16 = (int)(mb * 1024 * 1024 / 16);
        n16 = n16 / (8 * 64 * 8) * (8 * 64 * 8);
        uint4* buf; hipMalloc(&buf, (size_t)n16 * 16);
        std::vector<unsigned> h((size_t)n16 * 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
        hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const int passes = (int)(64.0 / mb) + 1;
        struct { const char* name; int mode; int stagger; } V[] = {{"lockstep", 0, 0}, {"stagger wg*8", 1, 8}, {"stagger wg*37", 1, 37}, {"stagger cu*8", 2, 8}, {"stagger cu*61", 2, 61}};
        for (auto& v : V) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(a);
                if (v.mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(wgs), dim3(512), 0, 0, buf, n16, passes, v.stagger, out);
                if (v.mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(wgs), dim3(512), 0, 0, buf, n16, passes, v.stagger, out);
                if (v.mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(wgs), dim3(512), 0, 0, buf, n16, passes, v.stagger, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            const double bytes = (double)wgs * passes * n16 * 16.0;
            printf("%5.1f MiB  %-14s %8.3f ms  %7.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz, %d CUs)\n", mb, v.name, best, bytes / best * 1e-9,
                   bytes / (best * 1e-3) / prop.multiProcessorCount / 2.1e9, prop.multiProcessorCount);
        }
        hipFree(buf);
    }
    return 0;
}
