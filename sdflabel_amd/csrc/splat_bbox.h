// Conservative screen box of a disc surfel (shared by the splat kernels and the fused surfel-forward kernel of project.hip).
#pragma once
#include "sdfr_common.h"
#ifndef SDFR_BOX_PAD
#define SDFR_BOX_PAD 0.25f
#endif

// Conservative pixel interval of the rays that can pass within rho of a point, along one image axis.
// A pixel is covered only if its ray passes closer than rho to the surfel centre, hence (projecting on the u-z plane)
// (pu - r pz)^2 < rho^2 (1 + r^2) with r = ray_u / ray_z.  Returns false if the interval is empty on screen.
__device__ __forceinline__ bool axis_range(float pu, float pz, float rho, float f, float c, int n, int& lo, int& hi) {
    lo = 0; hi = n - 1;
    const float A = pz * pz - rho * rho;
    if (!(A > 1e-9f) || !(f != 0.f)) return true;             // near the camera plane (or NaN): whole axis
    const float d2 = pu * pu + pz * pz;
    const float sq = rho * sqrtf(fmaxf(d2 - rho * rho, 0.f));
    const float r1 = (pu * pz - sq) / A, r2 = (pu * pz + sq) / A;
    float u1 = c + f * r1, u2 = c + f * r2;
    if (u1 > u2) { const float tmp = u1; u1 = u2; u2 = tmp; }
    // pad: the interval above is exact geometry (the tangent rays of the sphere of radius rho in the u-z plane, a superset of the disc's
    // footprint); floor / ceil below add up to a pixel per side; what the pad has to absorb is the float error of the quadratic, ~1e-6
    // relative.  0.25 px + 1e-3 relative (r01-r04 took 1.5 px: boxes of 215 px for discs that cover 53 -- four scan iterations per surfel in
    // the backward and twice the tile candidates in the forward; tests/test_gpu_splat.py::test_disc_screen_boxes_never_cut_a_covered_pixel
    // compares against box-free renders)
    const float pad = SDFR_BOX_PAD + 1e-3f * (fabsf(u1) + fabsf(u2));
    u1 -= pad; u2 += pad;
    if (isnan(u1) || isnan(u2)) return true;                  // undecidable: keep the whole axis
    if (u2 < 0.f || u1 > (float)(n - 1)) return false;       // entirely off screen
    lo = (int)fmaxf(floorf(u1), 0.f);
    hi = (int)fminf(ceilf(u2), (float)(n - 1));
    return lo <= hi;
}

// disc surfel (primitive 0): false = entirely off screen.  K row-major 3x3 of the crop; p = camera-frame centre.
__device__ __forceinline__ bool disc_bbox(const float* __restrict__ K, float px, float py, float pz, float diam, int W, int H, int& x0, int& y0,
                                          int& x1, int& y1) {
    x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1;
    const bool standard = (K[1] == 0.f) && (K[3] == 0.f) && (K[6] == 0.f) && (K[7] == 0.f) && (K[8] == 1.f);
    if (!standard) return true;
    if (!axis_range(px, pz, diam, K[0], K[2], W, x0, x1)) return false;
    if (!axis_range(py, pz, diam, K[4], K[5], H, y0, y1)) return false;
    return true;
}


// ---- binning of the screen boxes into 8x8 pixel tiles ---------------------------------------------------------------------------------
// Workspace of the splat forward pass (int32 words):  [B][cap][4] screen boxes, then per crop  tile_off[T + 2] | tile_list[SPL_LM * cap]
// with T = ceil(W/8) * ceil(H/8):  tile_off[t] .. tile_off[t+1] delimit tile t's entries of tile_list (surfel slots, in NO particular
// order: the consumer sorts them), tile_off[T] = total, tile_off[T + 1] = 1 if the lists are valid (0: too many tiles for the counters
// or more entries than SPL_LM * cap -- the splat kernel then scans all boxes per tile instead).
#define SPL_LM 32               // list capacity per crop = SPL_LM * cap entries
#define SPL_BIN_MAX_TILES 8192  // LDS counters of the binning pass (32 KiB)

__host__ __device__ __forceinline__ int64_t sdfr_splat_bin_stride(int cap, int W, int H) {
    return (int64_t)((W + 7) / 8) * ((H + 7) / 8) + 2 + (int64_t)SPL_LM * cap;
}

// One workgroup of NT threads bins `count` boxes of ONE crop (bbox -> tile_off / tile_list).  lds_cnt: T ints, lds_w: NT/64 + 1 ints.
// count -> exclusive scan -> fill, with LDS atomics (the per-tile order is whatever the atomics give; counts and offsets are exact).
template <int NT>
__device__ __forceinline__ void sdfr_bin_boxes(const int4* __restrict__ bbox, int count, int W, int H, int cap, int32_t* __restrict__ tile_off,
                                               int* lds_cnt, int* lds_w) {
    const int tid = threadIdx.x;
    const int tilesX = (W + 7) >> 3, tilesY = (H + 7) >> 3, T = tilesX * tilesY;
    int32_t* tile_list = tile_off + T + 2;
    const int list_cap = SPL_LM * cap;
    if (T > SPL_BIN_MAX_TILES) { if (tid == 0) { tile_off[T] = 0; tile_off[T + 1] = 0; } return; }
    for (int t = tid; t < T; t += NT) lds_cnt[t] = 0;
    __syncthreads();
    for (int s = tid; s < count; s += NT) {
        const int4 bb = bbox[s];
        if (bb.x > bb.z || bb.y > bb.w) continue;
        for (int ty = bb.y >> 3; ty <= (bb.w >> 3); ++ty)
            for (int tx = bb.x >> 3; tx <= (bb.z >> 3); ++tx) atomicAdd(&lds_cnt[ty * tilesX + tx], 1);
    }
    __syncthreads();
    // exclusive scan over the tiles: thread tid owns the contiguous tiles [tid*per, tid*per + per)
    const int per = (T + NT - 1) / NT;
    int local = 0;
    for (int k = 0; k < per; ++k) { const int t = tid * per + k; if (t < T) local += lds_cnt[t]; }
    int incl = local;
    const int lane = tid & 63, wv = tid >> 6;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) lds_w[wv] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < NT / 64; ++w) { const int c = lds_w[w]; lds_w[w] = run; run += c; }
        lds_w[NT / 64] = run;
    }
    __syncthreads();
    const int total = lds_w[NT / 64];
    int run = lds_w[wv] + incl - local;
    for (int k = 0; k < per; ++k) {
        const int t = tid * per + k;
        if (t < T) { const int c = lds_cnt[t]; tile_off[t] = run; lds_cnt[t] = run; run += c; }       // lds_cnt becomes the fill cursor
    }
    const bool ok = total <= list_cap;
    if (tid == 0) { tile_off[T] = total; tile_off[T + 1] = ok ? 1 : 0; }
    __syncthreads();
    if (!ok) return;
    for (int s = tid; s < count; s += NT) {
        const int4 bb = bbox[s];
        if (bb.x > bb.z || bb.y > bb.w) continue;
        for (int ty = bb.y >> 3; ty <= (bb.w >> 3); ++ty)
            for (int tx = bb.x >> 3; tx <= (bb.z >> 3); ++tx) tile_list[atomicAdd(&lds_cnt[ty * tilesX + tx], 1)] = s;
    }
}
