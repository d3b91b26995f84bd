// Sphere tracing of the DeepSDF level set: ray set-up and the per-step advance / early-termination compaction.
//
// NOT in the reference (its renderer splats surfels of a grid band); this is the render mode BASELINE.json's north_star
// describes literally -- "per-ray sphere-tracing loop and DeepSDF-MLP evaluation at each march step ... wavefront ballot for
// early-termination compaction" -- offered beside the faithful path, with no parity claim against the reference.
//   step:   x = o + lam d   ->   s = decoder(latent, x)  (sdfr_mlp_forward_counted on the ACTIVE rays only)   ->   lam += s / |d|
//   a ray leaves the active list when |s| < eps (hit: its lam is recorded per pixel) or when lam passes the far side of the object cube.
// Rays live in object space: p_cam = R p + t (the optimizer's pose, pipelines/optimizer.py:86-90), pixel ray r = K^-1 [x, y, 1]
// (primitives.py:203-208), so o = -R^T t, d = R^T r and lam is the camera-frame depth of the point (r_z = 1 for a pinhole K).
// The active list is compacted every step with one wave ballot + one atomic per wavefront (rays are independent: their order in the list
// does not matter), the count stays on the device, and the decoder launch of the next step reads it there: no host synchronisation.
#include "sdfr_common.h"
#include <float.h>

struct TraceRay { float ox, oy, oz, dx, dy, dz; };

__device__ __forceinline__ TraceRay trace_ray(const float* __restrict__ P, const float* __restrict__ Ki, float x, float y) {
    // pixel ray in the camera frame (same arithmetic as the splat's pixel_ray)
    const float rx = fmaf(Ki[1], y, Ki[0] * x) + Ki[2];
    const float ry = fmaf(Ki[4], y, Ki[3] * x) + Ki[5];
    const float rz = fmaf(Ki[7], y, Ki[6] * x) + Ki[8];
    TraceRay r;
    // d = R^T r, o = -R^T t   (P row-major 4x4, rotation in P[0..2], P[4..6], P[8..10], translation P[3], P[7], P[11])
    r.dx = P[0] * rx + P[4] * ry + P[8] * rz;
    r.dy = P[1] * rx + P[5] * ry + P[9] * rz;
    r.dz = P[2] * rx + P[6] * ry + P[10] * rz;
    r.ox = -(P[0] * P[3] + P[4] * P[7] + P[8] * P[11]);
    r.oy = -(P[1] * P[3] + P[5] * P[7] + P[9] * P[11]);
    r.oz = -(P[2] * P[3] + P[6] * P[7] + P[10] * P[11]);
    return r;
}

// append `keep` lanes to a list: one ballot + one atomic per wavefront; returns the slot of this lane (valid if keep)
__device__ __forceinline__ int trace_append(bool keep, int32_t* __restrict__ counter) {
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(counter, __popcll(bal));
    base = __shfl(base, 0, 64);
    return base + __popcll(bal & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void trace_write_row(float* __restrict__ row, const float* __restrict__ latn, int L, const TraceRay& r, float lam) {
    for (int c = 0; c < L; ++c) row[c] = latn[c];
    row[L] = r.ox + lam * r.dx; row[L + 1] = r.oy + lam * r.dy; row[L + 2] = r.oz + lam * r.dz;
}

// every pixel of every crop: slab test against the cube [-bound, bound]^3 the SDF is defined on; rays that hit it enter the active list
__global__ __launch_bounds__(256) void sdfr_trace_setup_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                              const float* __restrict__ latn, int L, int W, int H, float bound, float near,
                                                              int32_t* __restrict__ counters, int32_t* __restrict__ pix, float* __restrict__ lam,
                                                              float* __restrict__ far, float* __restrict__ inputs) {
    const int b = blockIdx.y;
    const int P_ = W * H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool active = false;
    TraceRay r = {};
    float l0 = 0.f, l1 = 0.f;
    if (p < P_) {
        r = trace_ray(pose + (int64_t)b * 16, Kinv + (int64_t)b * 9, (float)(p % W), (float)(p / W));
        l0 = near; l1 = FLT_MAX;
        const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
        active = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (fabsf(d[a]) < 1e-12f) { active = active && (fabsf(o[a]) <= bound); continue; }
            float ta = (-bound - o[a]) / d[a], tb = (bound - o[a]) / d[a];
            if (ta > tb) { const float t = ta; ta = tb; tb = t; }
            l0 = fmaxf(l0, ta); l1 = fminf(l1, tb);
        }
        active = active && (l0 < l1);
        far[(int64_t)b * P_ + p] = active ? l1 : 0.f;
    }
    const int slot = trace_append(active, counters);
    if (active) {
        pix[slot] = b * P_ + p;
        lam[slot] = l0;
        trace_write_row(inputs + (int64_t)slot * (L + 3), latn + (int64_t)b * L, L, r, l0);
    }
}

// one march step of every active ray (count on the device): advance by the decoder's value, retire hits and exits, compact the survivors
__global__ __launch_bounds__(256) void sdfr_trace_step_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                             const float* __restrict__ latn, int L, int W, int H, float eps, float relax,
                                                             const float* __restrict__ sdf, const int32_t* __restrict__ n_cur,
                                                             int32_t* __restrict__ n_next, int32_t* __restrict__ n_zero,
                                                             const int32_t* __restrict__ pix_in, const float* __restrict__ lam_in,
                                                             int32_t* __restrict__ pix_out, float* __restrict__ lam_out,
                                                             const float* __restrict__ far, float* __restrict__ inputs,
                                                             float* __restrict__ hit_lam, float* __restrict__ hit_sdf) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int n = *n_cur;
    if (s == 0) *n_zero = 0;                     // the counter of the step after next (three counters rotate)
    if (blockIdx.x * 256 >= n) return;
    bool keep = false;
    int gp = 0;
    float l2 = 0.f;
    TraceRay r = {};
    if (s < n) {
        gp = pix_in[s];
        const int P_ = W * H, b = gp / P_, p = gp - b * P_;
        r = trace_ray(pose + (int64_t)b * 16, Kinv + (int64_t)b * 9, (float)(p % W), (float)(p / W));
        const float v = sdf[s], l = lam_in[s];
        if (fabsf(v) < eps) {                    // on the surface: retire as a hit
            hit_lam[gp] = l;
            hit_sdf[gp] = v;
        } else {
            l2 = l + relax * v / sqrtf(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz);
            keep = (l2 < far[gp]) && (v == v);   // past the cube (or NaN): a miss
        }
    }
    const int slot = trace_append(keep, n_next);
    if (keep) {
        const int b = gp / (W * H);
        pix_out[slot] = gp;
        lam_out[slot] = l2;
        trace_write_row(inputs + (int64_t)slot * (L + 3), latn + (int64_t)b * L, L, r, l2);
    }
}

extern "C" int sdfr_trace_setup(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound, float near,
                                int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, void* stream) {
    SDFR_REQUIRE(pose && Kinv && latn && counters && pix && lam && far && inputs, "sdfr_trace_setup: NULL argument");
    SDFR_REQUIRE(L >= 0 && B > 0 && W > 0 && H > 0 && bound > 0.f, "sdfr_trace_setup: bad size");
    hipStream_t s = (hipStream_t)stream;
    SDFR_HIP_CHECK(hipMemsetAsync(counters, 0, 3 * sizeof(int32_t), s));
    hipLaunchKernelGGL(sdfr_trace_setup_kernel, dim3(sdfr_cdiv((int64_t)W * H, 256), B), dim3(256), 0, s, pose, Kinv, latn, L, W, H, bound, near,
                       counters, pix, lam, far, inputs);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_trace_step(const float* pose, const float* Kinv, const float* latn, int L, int W, int H, float eps, float relax,
                               const float* sdf, int32_t* counters, int step, int64_t n_max, const int32_t* pix_in, const float* lam_in,
                               int32_t* pix_out, float* lam_out, const float* far, float* inputs, float* hit_lam, float* hit_sdf, void* stream) {
    SDFR_REQUIRE(pose && Kinv && latn && sdf && counters && pix_in && lam_in && pix_out && lam_out && far && inputs && hit_lam && hit_sdf,
                 "sdfr_trace_step: NULL argument");
    SDFR_REQUIRE(step >= 0 && n_max >= 0, "sdfr_trace_step: bad size");
    if (n_max == 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_trace_step_kernel, dim3(sdfr_cdiv(n_max, 256)), dim3(256), 0, (hipStream_t)stream, pose, Kinv, latn, L, W, H, eps, relax,
                       sdf, counters + step % 3, counters + (step + 1) % 3, counters + (step + 2) % 3, pix_in, lam_in, pix_out, lam_out, far, inputs,
                       hit_lam, hit_sdf);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
