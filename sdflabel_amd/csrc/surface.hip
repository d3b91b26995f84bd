// Zero-isosurface band selection / projection and the DeepSDF input-gradient scatter (gfx950).
//
// Replaces Grid3D.get_surface_points (reference sdfrenderer/grid.py:43-71): band mask |sdf| < thr (:64),
// order-preserving compaction (masked_select, :65-66) done with wave ballots instead of a dense boolean mask,
// n_hat = n/||n|| (:57-58), p = x - sdf*n_hat (:61), nocs = (p+1)/2 (:67), and the autograd backward of (:61,:67).
// Compiled with -ffp-contract=off: every multiply/add below rounds separately, like the reference's ATen ops.
#include "sdfr_common.h"

// ---- band selection -------------------------------------------------------------------------------------------
// pass 1: per-256-row block counts; pass 2: block offset = sum of the preceding block counts of the same crop,
// rank inside the block from ballot/popcount prefix.  No inter-workgroup hand-off inside a launch.

__global__ __launch_bounds__(256) void sdfr_band_count_kernel(const float* __restrict__ sdf, int64_t G, float thr,
                                                             const float* __restrict__ thr_extra, int32_t* __restrict__ blockcnt,
                                                             const int32_t* __restrict__ skip) {
    const int b = blockIdx.y;
    if (skip && skip[b]) return;                     // this crop keeps its previous selection
    if (thr_extra) thr += thr_extra[b];
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool in = false;
    if (g < G) in = fabsf(sdf[(int64_t)b * G + g]) < thr;
    const unsigned long long bal = __ballot(in);
    __shared__ int wc[4];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[(int64_t)b * gridDim.x + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ __launch_bounds__(256) void sdfr_band_scatter_kernel(const float* __restrict__ sdf, int64_t G, float thr,
                                                               const float* __restrict__ thr_extra,
                                                               const int32_t* __restrict__ blockcnt, int32_t* __restrict__ idx,
                                                               int cap, int32_t* __restrict__ cnt, int32_t* __restrict__ slot,
                                                               const int32_t* __restrict__ skip, int32_t* __restrict__ over, int over_bit) {
    const int b = blockIdx.y;
    if (skip && skip[b]) return;
    if (thr_extra) thr += thr_extra[b];
    const int nblk = gridDim.x;
    const int tid = threadIdx.x;
    __shared__ int part[256];
    __shared__ int wc[4];
    // offset of this block = sum of counts of blocks [0, blockIdx.x) of crop b; the last block also needs the total
    int s = 0;
    for (int i = tid; i < (int)blockIdx.x; i += 256) s += blockcnt[(int64_t)b * nblk + i];
    part[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) part[tid] += part[tid + st];
        __syncthreads();
    }
    const int base = part[0];
    const int64_t g = (int64_t)blockIdx.x * 256 + tid;
    bool in = false;
    if (g < G) in = fabsf(sdf[(int64_t)b * G + g]) < thr;
    const unsigned long long bal = __ballot(in);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wc[wv] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += wc[w];
    const int rank = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
    if (g < G) {
        int sl = -1;
        if (in && rank < cap) {
            idx[(int64_t)b * cap + rank] = (int32_t)g;
            sl = rank;
        }
        if (slot) slot[(int64_t)b * G + g] = sl;
    }
    if (blockIdx.x == nblk - 1 && tid == 0) {
        const int total = base + wc[0] + wc[1] + wc[2] + wc[3];
        cnt[b] = total;
        if (over && total > cap) atomicOr(&over[b], over_bit);           // sticky: survives later selections that fit again (r06)
    }
}

// The same selection with ONE workgroup per crop (r06): count, offsets and scatter in a single launch -- for launches of a few crops, where the
// two-kernel form above is two launch latencies around microseconds of work (the per-annotation refinement: B = 1).  Rows are walked 1024 at
// a time with a running offset: the same ascending order, the same idx / slot / cnt.
__global__ __launch_bounds__(1024) void sdfr_band_select_crop_kernel(const float* __restrict__ sdf, int64_t G, float thr,
                                                                    const float* __restrict__ thr_extra, int32_t* __restrict__ idx, int cap,
                                                                    int32_t* __restrict__ cnt, int32_t* __restrict__ slot,
                                                                    const int32_t* __restrict__ skip, int32_t* __restrict__ over, int over_bit) {
    const int b = blockIdx.x;
    if (skip && skip[b]) return;
    if (thr_extra) thr += thr_extra[b];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int wc[2][16];
    int base = 0;
    int it = 0;
    for (int64_t g0 = 0; g0 < G; g0 += 1024, ++it) {
        const int64_t g = g0 + tid;
        bool in = false;
        if (g < G) in = fabsf(sdf[(int64_t)b * G + g]) < thr;
        const unsigned long long bal = __ballot(in);
        int* w = wc[it & 1];                                             // (two buffers: one barrier per round)
        if (lane == 0) w[wv] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int c = w[i]; woff += (i < wv) ? c : 0; tot += c; }
        const int rank = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
        if (g < G) {
            int sl = -1;
            if (in && rank < cap) { idx[(int64_t)b * cap + rank] = (int32_t)g; sl = rank; }
            if (slot) slot[(int64_t)b * G + g] = sl;
        }
        base += tot;
    }
    if (tid == 0) {
        cnt[b] = base;
        if (over && base > cap) atomicOr(&over[b], over_bit);
    }
}

extern "C" int sdfr_band_select_margin(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, int32_t* idx, int cap,
                                       int32_t* cnt, int32_t* slot, int32_t* scratch, void* stream);
static int band_select_impl(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, const int32_t* skip, int32_t* idx, int cap,
                            int32_t* cnt, int32_t* slot, int32_t* scratch, void* stream, int32_t* over = nullptr, int over_bit = 0);
extern "C" int sdfr_band_select(const float* sdf, int64_t G, int B, float thr, int32_t* idx, int cap, int32_t* cnt,
                                int32_t* slot, int32_t* scratch, void* stream) {
    return sdfr_band_select_margin(sdf, G, B, thr, nullptr, idx, cap, cnt, slot, scratch, stream);
}

// the same with a per-crop addition to the threshold (thr + thr_extra[b]; thr_extra may be NULL): the candidate selection of the
// two-stage evaluation, whose safety margin is kept per crop on the device (sdfr_prefilter_guard)
extern "C" int sdfr_band_select_margin(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, int32_t* idx, int cap,
                                       int32_t* cnt, int32_t* slot, int32_t* scratch, void* stream) {
    return band_select_impl(sdf, G, B, thr, thr_extra, nullptr, idx, cap, cnt, slot, scratch, stream);
}

// ... and with per-crop skip flags (device int32[B]): flagged crops keep idx / cnt / slot of their previous selection
extern "C" int sdfr_band_select_skip(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, const int32_t* skip, int32_t* idx,
                                     int cap, int32_t* cnt, int32_t* slot, int32_t* scratch, void* stream) {
    return band_select_impl(sdf, G, B, thr, thr_extra, skip, idx, cap, cnt, slot, scratch, stream);
}

// ... and with a STICKY truncation flag (r06): over[b] |= over_bit whenever crop b's selection holds more than `cap` rows.  cnt[b] is
// overwritten by every selection, so a band that overflows in iterations 5-40 of a graph-replayed refinement and fits again at the end would
// pass a check of the last count alone; the flag stays until the caller clears it (BatchRenderer.check_overflow / set_crops).  The reference
// has no capacity (grid.py:64-66).  Launches of up to SDFR_BAND_ONE_WG_CROPS crops WITH skip flags take the one-workgroup-per-crop kernel (one launch).
#define SDFR_BAND_ONE_WG_CROPS 8
extern "C" int sdfr_band_select_ex(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, const int32_t* skip, int32_t* idx,
                                   int cap, int32_t* cnt, int32_t* slot, int32_t* scratch, int32_t* over, int over_bit, void* stream) {
    return band_select_impl(sdf, G, B, thr, thr_extra, skip, idx, cap, cnt, slot, scratch, stream, over, over_bit);
}

static int band_select_impl(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, const int32_t* skip, int32_t* idx, int cap,
                            int32_t* cnt, int32_t* slot, int32_t* scratch, void* stream, int32_t* over, int over_bit) {
    SDFR_REQUIRE(sdf && idx && cnt && scratch, "sdfr_band_select: NULL argument");
    SDFR_REQUIRE(G >= 0 && B >= 0 && cap >= 0, "sdfr_band_select: negative size");
    if (B == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    if (G == 0) { SDFR_HIP_CHECK(sdfr_zero_async(cnt, sizeof(int32_t) * B, s)); return SDFR_OK; }
    // (only where the selection is skipped on most steps -- the reuse modes' per-crop flags: one launch that usually has nothing to do.  A
    // selection that really scans its 64 000 rows with one workgroup takes 47 us against 5 + 6 us for the two-kernel form: r06's first builds took
    // this path in the headline step too and paid 36 us per step for it, profiles/r06_kernel_stats.csv)
    if (over && skip && B <= SDFR_BAND_ONE_WG_CROPS) {
        hipLaunchKernelGGL(sdfr_band_select_crop_kernel, dim3(B), dim3(1024), 0, s, sdf, G, thr, thr_extra, idx, cap, cnt, slot, skip, over, over_bit);
        SDFR_LAUNCH_CHECK();
        return SDFR_OK;
    }
    dim3 grid(sdfr_cdiv(G, 256), B);
    hipLaunchKernelGGL(sdfr_band_count_kernel, grid, dim3(256), 0, s, sdf, G, thr, thr_extra, scratch, skip);
    SDFR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sdfr_band_scatter_kernel, grid, dim3(256), 0, s, sdf, G, thr, thr_extra, scratch, idx, cap, cnt, slot, skip, over, over_bit);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- two-stage band selection: exact values patched into a prefiltered grid, Jacobian rows re-ordered ---------------------
// A cheap (half-operand) decoder pass over the whole grid followed by band_select with a safety margin yields CANDIDATE rows; the
// exact decoder pass runs on the candidates only (sdfr_mlp_jacobian without masks gives their float32 sdf and Jacobian).
// sdfr_scatter_values writes those exact sdf values back into the grid array, so that the ordinary band_select on it returns the
// exact band; sdfr_gather_rows then brings the candidates' Jacobian rows into band order.

__global__ __launch_bounds__(256) void sdfr_scatter_values_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                 const int32_t* __restrict__ idx, int64_t G, int cap,
                                                                 const int32_t* __restrict__ cnt) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= sdfr_count(cnt, b, cap)) return;
    const int64_t e = (int64_t)b * cap + s;
    dst[(int64_t)b * G + idx[e]] = src[e];
}

extern "C" int sdfr_scatter_values(float* dst, const float* src, const int32_t* idx, int64_t G, int B, int cap, const int32_t* cnt,
                                   void* stream) {
    SDFR_REQUIRE(dst && src && idx, "sdfr_scatter_values: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_scatter_values_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, (hipStream_t)stream, dst, src, idx, G, cap,
                       cnt);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// sdfr_scatter_values + the run-time guard of the two-stage evaluation, one workgroup per crop.  The exclusion test of the half pass is only
// valid while its error stays below the margin; the error is observable exactly where it matters -- at the candidates, whose exact values
// have just been computed.  dev = max |half-pass value - exact value| over the crop's candidates:
//   dev > margin / 2   "soft" violation: counted, and the crop's margin grows to 4 * dev for the following selections
//   dev >= margin      "hard" violation: counted separately -- a row may have been excluded wrongly in THIS step
// Everything stays on the device (no host synchronisation; graph-capturable); the host reads the counters when convenient.
__global__ __launch_bounds__(1024) void sdfr_prefilter_guard_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                   const int32_t* __restrict__ idx, int64_t G, int cap,
                                                                   const int32_t* __restrict__ cnt, float* __restrict__ margin,
                                                                   float* __restrict__ max_dev, int32_t* __restrict__ violations,
                                                                   const int32_t* __restrict__ reused) {
    const int b = blockIdx.x;
    const int n = sdfr_count(cnt, b, cap);
    float dev = 0.f;
    for (int s = threadIdx.x; s < n; s += 1024) {
        const int64_t e = (int64_t)b * cap + s;
        const int64_t r = (int64_t)b * G + idx[e];
        const float v = src[e];
        dev = fmaxf(dev, fabsf(dst[r] - v));
        dst[r] = v;
    }
    for (int o = 32; o > 0; o >>= 1) dev = fmaxf(dev, __shfl_xor(dev, o, 64));
    __shared__ float wmax[16];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = dev;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) dev = fmaxf(dev, wmax[w]);
        if (reused && reused[b]) return;                  // candidate set reused: the old values were exact ones, not the half pass's
        const float m = margin[b];
        max_dev[b] = dev;
        if (!(dev <= 0.5f * m)) {                         // also catches NaN
            violations[2 * b] += 1;
            if (!(dev < m)) violations[2 * b + 1] += 1;
            margin[b] = fmaxf(m, 4.f * dev);
        }
    }
}

// ---- audit of the two-stage evaluation (r04) ------------------------------------------------------------------------------
// The guard above sees the half pass's error only at the CANDIDATES.  A row outside them that the half pass misplaced by more than the margin
// (so that it belongs to the band but was never proposed) is invisible to it.  The audit closes that: every step, the non-candidate rows of
// ONE slice of the grid -- the contiguous rows [ph * ceil(G / stride), (ph + 1) * ceil(G / stride)), ph = phase mod stride: a rotating
// 1/stride of the grid, every row once per `stride` steps -- are evaluated with
// the exact float32 decoder (sdfr_mlp_forward_counted on the compact row list written here) and judged by sdfr_prefilter_audit_check:
//   |exact| < thr on a non-candidate row  ->  a band row WAS excluded in this step: hard violation (violations[2b+1]), the result is refused
//                                              like a candidate-side hard violation (BatchRenderer.check_overflow raises);
//   |half - exact| (crops that ran the half pass this step)  ->  audit_dev[b] = the largest seen since the last reset, for the report.
// Everything on the device: the phase is a device counter advanced by the check kernel (graph-capturable, no host state).
__global__ __launch_bounds__(256) void sdfr_prefilter_audit_select_kernel(const float* __restrict__ inputs, const int32_t* __restrict__ cslot,
                                                                         int64_t G, int NI, int stride, const int32_t* __restrict__ phase,
                                                                         float* __restrict__ rows, int32_t* __restrict__ src,
                                                                         int32_t* __restrict__ n_audit, int cap_rows) {
    const int b = blockIdx.y;
    const int ph = ((*phase) % stride + stride) % stride;
    // slice `ph` of the grid: the contiguous rows [ph * per, (ph + 1) * per), per = ceil(G / stride) (r05; r04 took the residue class
    // g = ph (mod stride), whose reads of cslot and of the input rows were `stride` rows apart: 61 us at 64 crops, now coalesced)
    const int64_t per = (G + stride - 1) / stride;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;          // k-th row of the slice
    const int64_t g = (int64_t)ph * per + k;
    const bool take = k < per && g < G && cslot[(int64_t)b * G + g] < 0;
    const unsigned long long bal = __ballot(take);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(n_audit, __popcll(bal));
    base = __shfl(base, 0, 64);
    const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (take && pos < cap_rows) {
        const float* in = inputs + ((int64_t)b * G + g) * NI;
        float* o = rows + (int64_t)pos * NI;
        for (int c = 0; c < NI; ++c) o[c] = in[c];
        src[pos] = (int)((int64_t)b * G + g);
    }
}

__global__ __launch_bounds__(256) void sdfr_prefilter_audit_check_kernel(const float* __restrict__ sdf_grid, const float* __restrict__ sdf_exact,
                                                                        const int32_t* __restrict__ src, const int32_t* __restrict__ n_audit,
                                                                        int cap_rows, int64_t G, float thr, const int32_t* __restrict__ reused,
                                                                        float* __restrict__ audit_dev, int32_t* __restrict__ violations,
                                                                        int32_t* __restrict__ phase) {
    const int n = min(*n_audit, cap_rows);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int r = src[i];
        const int b = (int)(r / G);
        const float ex = sdf_exact[i];
        if (!(fabsf(ex) >= thr)) atomicAdd(&violations[2 * b + 1], 1);                 // a band row outside the candidates (or NaN): missed
        if (!(reused && reused[b])) {
            const float dev = fabsf(sdf_grid[r] - ex);                                  // (non-negative floats order like their bit patterns)
            // (a plain read first: the running maximum settles after a few steps and most rows then need no atomic at all -- 244 000 atomics
            // on 64 addresses cost 115 us per step at 64 crops)
            if (dev > audit_dev[b]) atomicMax(reinterpret_cast<int*>(audit_dev) + b, __float_as_int(dev));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // (every block of the select kernel has read the phase: this launch follows it in stream order)
        *phase = *phase + 1;
    }
}

extern "C" int sdfr_prefilter_audit_select(const float* inputs, const int32_t* cslot, int64_t G, int n_inputs, int B, int stride, const int32_t* phase,
                                           float* rows, int32_t* src, int32_t* n_audit, int cap_rows, void* stream) {
    SDFR_REQUIRE(inputs && cslot && phase && rows && src && n_audit, "sdfr_prefilter_audit_select: NULL argument");
    SDFR_REQUIRE(G > 0 && n_inputs > 0 && stride > 0 && cap_rows > 0, "sdfr_prefilter_audit_select: bad size");
    if (B <= 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    SDFR_HIP_CHECK(sdfr_zero_async(n_audit, sizeof(int32_t), s));
    const int per = sdfr_cdiv(G, stride);
    hipLaunchKernelGGL(sdfr_prefilter_audit_select_kernel, dim3(sdfr_cdiv(per, 256), B), dim3(256), 0, s, inputs, cslot, G, n_inputs, stride, phase,
                       rows, src, n_audit, cap_rows);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_prefilter_audit_check(const float* sdf_grid, const float* sdf_exact, const int32_t* src, const int32_t* n_audit, int cap_rows,
                                          int64_t G, int B, float thr, const int32_t* reused, float* audit_dev, int32_t* violations,
                                          int32_t* phase, void* stream) {
    SDFR_REQUIRE(sdf_grid && sdf_exact && src && n_audit && audit_dev && violations && phase, "sdfr_prefilter_audit_check: NULL argument");
    SDFR_REQUIRE(G > 0 && cap_rows > 0, "sdfr_prefilter_audit_check: bad size");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_prefilter_audit_check_kernel, dim3(sdfr_cdiv(cap_rows, 256)), dim3(256), 0, (hipStream_t)stream, sdf_grid, sdf_exact, src,
                       n_audit, cap_rows, G, thr, reused, audit_dev, violations, phase);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// Plan of the two-stage evaluation for this step, per crop (one thread each): may the candidate set of the last half pass be reused?
// A row outside it had |half sdf| >= thr + margin at the latent z0 of that pass, so its exact |sdf| at the current latent z1 is at least
// thr + margin - dev - lip * |z1 - z0|  (dev: the half pass's deviation, lip: Lipschitz constant of the decoder in the normalised latent,
// calibrated by the caller).  Reuse while  lip * |z1 - z0| <= margin / 4  and  dev <= margin / 2  (then the row stays outside the band with
// a quarter of the margin to spare) and at most max_reuse steps in a row.  reuse[b] = 1: skip the half pass and the candidate selection of
// crop b (sdfr_mlp_forward_f16_skip, sdfr_band_select_skip); otherwise lat_ref[b] = the current normalised latent.
__global__ void sdfr_prefilter_plan_kernel(const float* __restrict__ inputs, int64_t G, int NI, int L, int B, float lip,
                                           const float* __restrict__ margin, const float* __restrict__ max_dev, float* __restrict__ lat_ref,
                                           int32_t* __restrict__ age, int max_reuse, int32_t* __restrict__ reuse, int32_t* __restrict__ n_full) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* z = inputs + (int64_t)b * G * NI;             // the latent columns of the crop's first input row
    float d2 = 0.f;
    for (int c = 0; c < L; ++c) { const float d = z[c] - lat_ref[b * L + c]; d2 += d * d; }
    const bool ok = age[b] > 0 && age[b] <= max_reuse && lip * sqrtf(d2) <= 0.25f * margin[b] && max_dev[b] <= 0.5f * margin[b];
    reuse[b] = ok ? 1 : 0;
    if (ok) age[b] += 1;
    else {
        age[b] = 1;
        for (int c = 0; c < L; ++c) lat_ref[b * L + c] = z[c];
        if (n_full) n_full[b] += 1;                            // full-grid passes of this crop since the caller zeroed the counter
    }
}

extern "C" int sdfr_prefilter_plan(const float* inputs, int64_t G, int n_inputs, int L, int B, float lip, const float* margin,
                                   const float* max_dev, float* lat_ref, int32_t* age, int max_reuse, int32_t* reuse, int32_t* n_full, void* stream) {
    SDFR_REQUIRE(inputs && margin && max_dev && lat_ref && age && reuse && G > 0 && L >= 0 && n_inputs >= L, "sdfr_prefilter_plan: bad argument");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_prefilter_plan_kernel, dim3(sdfr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, inputs, G, n_inputs, L, B, lip, margin,
                       max_dev, lat_ref, age, max_reuse, reuse, n_full);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_prefilter_guard2(float* sdf_grid, const float* sdf_exact, const int32_t* idx, int64_t G, int B, int cap, const int32_t* cnt,
                                     float* margin, float* max_dev, int32_t* violations, const int32_t* reused, void* stream);
extern "C" int sdfr_prefilter_guard(float* sdf_grid, const float* sdf_exact, const int32_t* idx, int64_t G, int B, int cap,
                                    const int32_t* cnt, float* margin, float* max_dev, int32_t* violations, void* stream) {
    return sdfr_prefilter_guard2(sdf_grid, sdf_exact, idx, G, B, cap, cnt, margin, max_dev, violations, nullptr, stream);
}

// ... with the plan's reuse flags: crops that reused their candidate set are patched but not judged (their old values were exact ones)
extern "C" int sdfr_prefilter_guard2(float* sdf_grid, const float* sdf_exact, const int32_t* idx, int64_t G, int B, int cap, const int32_t* cnt,
                                     float* margin, float* max_dev, int32_t* violations, const int32_t* reused, void* stream) {
    SDFR_REQUIRE(sdf_grid && sdf_exact && idx && margin && max_dev && violations, "sdfr_prefilter_guard: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_prefilter_guard_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, sdf_grid, sdf_exact, idx, G, cap, cnt, margin,
                       max_dev, violations, reused);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- float16 candidate reuse (r05) ------------------------------------------------------------------------------------------------------
// The float16 decoder mode evaluates the whole grid every iteration although only the latent moves, and by ~1e-6 per iteration
// (pipelines/optimizer.py:34-38: lr 3e-5).  With decoder.candidate_reuse the band comes from the CANDIDATE rows (|half sdf| < thr + margin at
// the last full pass) alone: their input rows are gathered into a ragged [B][stride] array (sdfr_candidate_rows), evaluated with the SAME
// half kernel (sdfr_mlp_forward_f16_ragged: same bits per row as the full-grid launch), scattered back into the grid array
// (sdfr_scatter_values) and band-selected there as ever; sdfr_candidate_band_map turns the band's grid rows into positions of the candidate
// array, where the mask-fed half Jacobian finds its masks.  Valid while no row outside the candidates can have entered the band:
// sdfr_prefilter_plan with a PROVEN Lipschitz bound of the decoder in the latent (Decoder.latent_lipschitz_bound) decides per crop on the
// device, and the rotating audit (sdfr_prefilter_audit_*) re-evaluates 1 / stride of the other rows every step with the same kernel.
__global__ __launch_bounds__(256) void sdfr_candidate_rows_kernel(const float* __restrict__ inputs, int64_t G, int NI, const int32_t* __restrict__ cidx,
                                                                 int stride, const int32_t* __restrict__ ccnt, float* __restrict__ rows) {
    const int b = blockIdx.y;
    const int n = sdfr_count(ccnt, b, stride);
    const int n_pad = min(stride, (n + 127) / 128 * 128);              // the last 128-row tile is computed whole: finite rows up to its end
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int s = (int)(t / NI), c = (int)(t - (int64_t)s * NI);
    if (s >= n_pad) return;
    const int64_t g = s < n ? (int64_t)cidx[(int64_t)b * stride + s] : 0;   // padding rows: the crop's grid row 0
    rows[((int64_t)b * stride + s) * NI + c] = inputs[((int64_t)b * G + g) * NI + c];
}

extern "C" int sdfr_candidate_rows(const float* inputs, int64_t G, int n_inputs, int B, const int32_t* cidx, int stride, const int32_t* ccnt,
                                   float* rows, void* stream) {
    SDFR_REQUIRE(inputs && cidx && ccnt && rows && G > 0 && n_inputs > 0, "sdfr_candidate_rows: bad argument");
    SDFR_REQUIRE(stride >= 0 && stride % 128 == 0, "sdfr_candidate_rows: stride %d is not a multiple of 128", stride);
    if (B <= 0 || stride == 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_candidate_rows_kernel, dim3(sdfr_cdiv((int64_t)stride * n_inputs, 256), B), dim3(256), 0, (hipStream_t)stream, inputs, G,
                       n_inputs, cidx, stride, ccnt, rows);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// pos[b][e] = cslot[b*G + idx[b][e]] for e < cnt[b]: the band rows' positions in their crop's candidate array.  A band row that is no
// candidate (cannot happen while the candidate set is valid: its stale grid value is >= thr + margin) maps to position 0 and counts a hard
// violation (violations[2b+1]), which check_overflow() refuses.
__global__ __launch_bounds__(256) void sdfr_candidate_band_map_kernel(const int32_t* __restrict__ idx, int cap, const int32_t* __restrict__ cnt,
                                                                     const int32_t* __restrict__ cslot, int64_t G, int stride,
                                                                     int32_t* __restrict__ pos, int32_t* __restrict__ violations) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= sdfr_count(cnt, b, cap)) return;
    int sl = cslot[(int64_t)b * G + idx[(int64_t)b * cap + e]];
    if (sl < 0 || sl >= stride) {
        sl = 0;
        if (violations) atomicAdd(&violations[2 * b + 1], 1);
    }
    pos[(int64_t)b * cap + e] = sl;
}

extern "C" int sdfr_candidate_band_map(const int32_t* idx, int cap, const int32_t* cnt, const int32_t* cslot, int64_t G, int B, int stride,
                                       int32_t* pos, int32_t* violations, void* stream) {
    SDFR_REQUIRE(idx && cslot && pos && G > 0, "sdfr_candidate_band_map: bad argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_candidate_band_map_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, (hipStream_t)stream, idx, cap, cnt, cslot, G, stride,
                       pos, violations);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// sdfr_scatter_values + sdfr_band_select + sdfr_candidate_band_map in ONE launch (r06), one workgroup per crop: the band of a crop whose
// candidate set is valid lies INSIDE the candidates (a row outside them keeps its full-pass value, >= thr + margin by selection, in the grid
// array -- never a band row), and the candidates are listed in ascending grid-row order, so the order-preserving compaction of the candidate
// values IS the band list the grid-wide selection returns: idx[b][e] = cidx of the e-th candidate with |value| < thr, pos[b][e] = its position
// in the candidate array; the values are also written into the grid array (the downstream kernels read sdf[b*G + idx]).  Four launches that
// scan B x G rows become one that scans B x ~3 000.  over[b] |= 1 if the band exceeds cap, |= 2 if the candidates exceeded their stride.
__global__ __launch_bounds__(1024) void sdfr_candidate_band_kernel(float* __restrict__ sdf_grid, const float* __restrict__ csdf,
                                                                  const int32_t* __restrict__ cidx, int64_t G, int stride,
                                                                  const int32_t* __restrict__ ccnt, float thr, int32_t* __restrict__ idx, int cap,
                                                                  int32_t* __restrict__ cnt, int32_t* __restrict__ pos, int32_t* __restrict__ over) {
    const int b = blockIdx.x;
    const int n_raw = ccnt[b];
    const int n = n_raw < stride ? n_raw : stride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int wc[2][16];
    int base = 0, it = 0;
    for (int s0 = 0; s0 < n; s0 += 1024, ++it) {
        const int s = s0 + tid;
        bool in = false;
        int g = 0;
        if (s < n) {
            const float v = csdf[(int64_t)b * stride + s];
            g = cidx[(int64_t)b * stride + s];
            sdf_grid[(int64_t)b * G + g] = v;
            in = fabsf(v) < thr;
        }
        const unsigned long long bal = __ballot(in);
        int* w = wc[it & 1];
        if (lane == 0) w[wv] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int c = w[i]; woff += (i < wv) ? c : 0; tot += c; }
        const int rank = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
        if (in && rank < cap) { idx[(int64_t)b * cap + rank] = g; pos[(int64_t)b * cap + rank] = s; }
        base += tot;
    }
    if (tid == 0) {
        cnt[b] = base;
        if (over) {
            const int f = (base > cap ? 1 : 0) | (n_raw > stride ? 2 : 0);
            if (f) atomicOr(&over[b], f);
        }
    }
}

extern "C" int sdfr_candidate_band(float* sdf_grid, const float* csdf, const int32_t* cidx, int64_t G, int B, int stride, const int32_t* ccnt,
                                   float thr, int32_t* idx, int cap, int32_t* cnt, int32_t* pos, int32_t* over, void* stream) {
    SDFR_REQUIRE(sdf_grid && csdf && cidx && ccnt && idx && cnt && pos && G > 0 && stride >= 0 && cap >= 0, "sdfr_candidate_band: bad argument");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_candidate_band_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, sdf_grid, csdf, cidx, G, stride, ccnt, thr, idx, cap,
                       cnt, pos, over);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// out[b][e][:] = src[b][slot[b*G + idx[b][e]]][:]  for e < cnt[b]  (ncol floats per row; rows whose slot is negative become zero)
__global__ __launch_bounds__(256) void sdfr_gather_rows_kernel(float* __restrict__ out, const float* __restrict__ src, int ncol,
                                                              const int32_t* __restrict__ idx, const int32_t* __restrict__ slot,
                                                              int64_t G, int cap, int src_cap, const int32_t* __restrict__ cnt) {
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int e = (int)(t / ncol), c = (int)(t - (int64_t)e * ncol);
    if (e >= sdfr_count(cnt, b, cap)) return;
    const int sl = slot[(int64_t)b * G + idx[(int64_t)b * cap + e]];
    out[((int64_t)b * cap + e) * ncol + c] = (sl >= 0 && sl < src_cap) ? src[((int64_t)b * src_cap + sl) * ncol + c] : 0.f;
}

extern "C" int sdfr_gather_rows(float* out, const float* src, int ncol, const int32_t* idx, const int32_t* slot, int64_t G, int B,
                                int cap, int src_cap, const int32_t* cnt, void* stream) {
    SDFR_REQUIRE(out && src && idx && slot && ncol > 0, "sdfr_gather_rows: bad argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_gather_rows_kernel, dim3(sdfr_cdiv((int64_t)cap * ncol, 256), B), dim3(256), 0, (hipStream_t)stream, out, src,
                       ncol, idx, slot, G, cap, src_cap, cnt);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- projection onto the zero level set -----------------------------------------------------------------------

__global__ __launch_bounds__(256) void sdfr_surface_project_kernel(const float* __restrict__ xyz, int xyz_stride,
                                                                  const float* __restrict__ sdf, int64_t G,
                                                                  const int32_t* __restrict__ idx, int cap,
                                                                  const int32_t* __restrict__ cnt, const float* __restrict__ J,
                                                                  int Jstride, int Joff, float* __restrict__ points,
                                                                  float* __restrict__ nocs, float* __restrict__ normals) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= sdfr_count(cnt, b, cap)) return;
    const int64_t e = (int64_t)b * cap + s;
    const int64_t r = (int64_t)b * G + idx[e];
    const float* x = xyz + r * xyz_stride;
    const float* n = J + e * Jstride + Joff;
    const float nx = n[0], ny = n[1], nz = n[2];
    const float nrm = sqrtf(nx * nx + ny * ny + nz * nz);          // grid.py:57
    const float hx = nx / nrm, hy = ny / nrm, hz = nz / nrm;       // grid.py:58
    const float sd = sdf[r];
    const float px = x[0] - sd * hx, py = x[1] - sd * hy, pz = x[2] - sd * hz;   // grid.py:61
    points[e * 3 + 0] = px; points[e * 3 + 1] = py; points[e * 3 + 2] = pz;
    normals[e * 3 + 0] = hx; normals[e * 3 + 1] = hy; normals[e * 3 + 2] = hz;
    if (nocs) {
        nocs[e * 3 + 0] = (px + 1.f) / 2.f; nocs[e * 3 + 1] = (py + 1.f) / 2.f; nocs[e * 3 + 2] = (pz + 1.f) / 2.f;   // grid.py:67
    }
}

extern "C" int sdfr_surface_project(const float* xyz, int xyz_stride, const float* sdf, int64_t G, int B,
                                    const int32_t* idx, int cap, const int32_t* cnt, const float* J, int Jstride, int Joff,
                                    float* points, float* nocs, float* normals, void* stream) {
    SDFR_REQUIRE(xyz && sdf && idx && J && points && normals, "sdfr_surface_project: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    dim3 grid(sdfr_cdiv(cap, 256), B);
    hipLaunchKernelGGL(sdfr_surface_project_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, xyz_stride, sdf, G, idx, cap,
                       cnt, J, Jstride, Joff, points, nocs, normals);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

__global__ __launch_bounds__(256) void sdfr_surface_project_bwd_kernel(const float* __restrict__ g_points,
                                                                      const float* __restrict__ g_nocs,
                                                                      const float* __restrict__ normals, int64_t G,
                                                                      const int32_t* __restrict__ idx, int cap,
                                                                      const int32_t* __restrict__ cnt, float* __restrict__ g_sdf,
                                                                      float* __restrict__ g_xyz) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= sdfr_count(cnt, b, cap)) return;
    const int64_t e = (int64_t)b * cap + s;
    const int64_t r = (int64_t)b * G + idx[e];
    float gx = g_points[e * 3 + 0], gy = g_points[e * 3 + 1], gz = g_points[e * 3 + 2];
    if (g_nocs) { gx += g_nocs[e * 3 + 0] / 2.f; gy += g_nocs[e * 3 + 1] / 2.f; gz += g_nocs[e * 3 + 2] / 2.f; }
    const float hx = normals[e * 3 + 0], hy = normals[e * 3 + 1], hz = normals[e * 3 + 2];
    g_sdf[r] = -(gx * hx + gy * hy + gz * hz);
    if (g_xyz) { g_xyz[r * 3 + 0] = gx; g_xyz[r * 3 + 1] = gy; g_xyz[r * 3 + 2] = gz; }
}

extern "C" int sdfr_surface_project_bwd(const float* g_points, const float* g_nocs, const float* normals, int64_t G, int B,
                                        const int32_t* idx, int cap, const int32_t* cnt, float* g_sdf, float* g_xyz,
                                        void* stream) {
    SDFR_REQUIRE(g_points && normals && idx && g_sdf, "sdfr_surface_project_bwd: NULL argument");
    if (B <= 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    SDFR_HIP_CHECK(sdfr_zero_async(g_sdf, sizeof(float) * (size_t)B * G, s));
    if (g_xyz) SDFR_HIP_CHECK(sdfr_zero_async(g_xyz, sizeof(float) * (size_t)B * G * 3, s));
    if (cap <= 0) return SDFR_OK;
    dim3 grid(sdfr_cdiv(cap, 256), B);
    hipLaunchKernelGGL(sdfr_surface_project_bwd_kernel, grid, dim3(256), 0, s, g_points, g_nocs, normals, G, idx, cap, cnt,
                       g_sdf, g_xyz);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- DeepSDF backward through the cached band Jacobian --------------------------------------------------------

__global__ __launch_bounds__(256) void sdfr_sdf_input_grad_kernel(const float* __restrict__ g_sdf, const int32_t* __restrict__ slot,
                                                                 const float* __restrict__ J, int NI, int64_t G, int cap,
                                                                 float* __restrict__ g_inputs, int32_t* __restrict__ n_uncached) {
    const int b = blockIdx.y;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool miss = false;
    if (g < G) {
        const int64_t r = (int64_t)b * G + g;
        const float gs = g_sdf[r];
        const int sl = slot[r];
        float* out = g_inputs + r * NI;
        if (sl >= 0) {
            const float* j = J + ((int64_t)b * cap + sl) * NI;
            for (int c = 0; c < NI; ++c) out[c] = gs * j[c];
        } else {
            for (int c = 0; c < NI; ++c) out[c] = 0.f;
            miss = (gs != 0.f);
        }
    }
    const unsigned long long bal = __ballot(miss);
    if (n_uncached && bal && (threadIdx.x & 63) == 0) atomicAdd(n_uncached, (int)__popcll(bal));
}

extern "C" int sdfr_sdf_input_grad(const float* g_sdf, const int32_t* slot, const float* J, int n_inputs, int64_t G, int B,
                                   int cap, float* g_inputs, int32_t* n_uncached, void* stream) {
    SDFR_REQUIRE(g_sdf && slot && J && g_inputs, "sdfr_sdf_input_grad: NULL argument");
    if (B <= 0 || G <= 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    if (n_uncached) SDFR_HIP_CHECK(sdfr_zero_async(n_uncached, sizeof(int32_t), s));
    dim3 grid(sdfr_cdiv(G, 256), B);
    hipLaunchKernelGGL(sdfr_sdf_input_grad_kernel, grid, dim3(256), 0, s, g_sdf, slot, J, n_inputs, G, cap, g_inputs, n_uncached);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
