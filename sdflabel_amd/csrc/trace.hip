// Sphere tracing of the DeepSDF level set: ray set-up and the per-step advance / early-termination compaction.
//
// NOT in the reference (its renderer splats surfels of a grid band); this is the render mode BASELINE.json's north_star
// describes literally -- "per-ray sphere-tracing loop and DeepSDF-MLP evaluation at each march step ... wavefront ballot for
// early-termination compaction" -- offered beside the faithful path, with no parity claim against the reference.
//   step:   x = o + lam d   ->   s = decoder(latent, x)  (the decoder kernels on the ACTIVE rays only)   ->   lam += s / |d|
//   a ray leaves the active list when |s| < eps (hit: its lam is recorded per pixel) or when lam passes the far side of the object cube.
// Rays live in object space: p_cam = R p + t (the optimizer's pose, pipelines/optimizer.py:86-90), pixel ray r = K^-1 [x, y, 1]
// (primitives.py:203-208), so o = -R^T t, d = R^T r and lam is the camera-frame depth of the point (r_z = 1 for a pinhole K).
// The active list is compacted every step with one wave ballot + one atomic per wavefront (rays are independent: their order in the list
// does not matter), the count stays on the device, and the decoder launch of the next step reads it there: no host synchronisation.
// Ray state (float4 per active ray): lam = the next sample, rho = |s| of the previous accepted sample, q = ratio of the last two radii
// (clamped to [0.5, 1]) -- rho and q feed the speculative passes of the looping tail (mlp_kernel.h MODE 4: K samples per ray and pass,
// p_j = p_{j-1} + sigma q^j rho / |d|, accepted while each lies inside the previous one's safe sphere).
#include "mlp_kernel.h"
#include <float.h>

struct TraceRay { float ox, oy, oz, dx, dy, dz; };

// Image extents of the crops of a launch.  Dense: every crop W x H pixels, pixel slot PS = W H, cone slots ncap = its cone count.  Ragged (r04;
// wh != NULL): crop b is W_b = wh[2b] x H_b = wh[2b+1] pixels inside a slot of PS pixels (global pixel id gp = b PS + y W_b + x) and owns ncap cone
// slots -- every crop of a batch its own size and intrinsics, read on the device (the reference pipeline's crops: utils/refinement.py:586-609).
struct TraceDims { const int32_t* wh; int W, H, PS, ncap; };
__device__ __forceinline__ void trace_dims(const TraceDims& D, int b, int& W, int& H) {
    W = D.W; H = D.H;
    if (D.wh) {
        W = D.wh[2 * b]; H = D.wh[2 * b + 1];
        if (W < 1 || H < 1 || (int64_t)W * H > (int64_t)D.PS) { W = 0; H = 0; }        // outside the contract: an empty crop, not an out-of-bounds write
    }
}

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

// One plain sample of a ray: st = (lam, rho, q, -), v = decoder value at lam, dn = |d|.  Returns 1 hit (st.x = the hit's lam), 0 keep marching
// (st advanced), -1 miss (past the cube's far side, or NaN).  The K = 1 case of the march's step rule (oracle/sdf_oracle.py::sphere_trace);
// the looping tail of the decoder kernel (mlp_kernel.h MODE 4) applies the same rule to the accepted prefix of its K samples.
__device__ __forceinline__ int trace_advance(float4& st, float v, float dn, float eps, float far, float qmax) {
    const float r = fabsf(v);
    if (r < eps) return 1;
    const float q = (st.y > 0.f) ? fminf(fmaxf(r / st.y, 0.5f), qmax) : 1.f;
    const float l2 = st.x + v / dn;
    st.y = r;
    st.z = q;
    if (!(l2 < far) || !(v == v)) return -1;
    st.x = l2;
    return 0;
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

// the cone march evaluates the decoder on a block's CENTRE ray anywhere between the block's nearest entry and farthest exit, so the centre
// point x can lie outside the cube [-bound, bound]^3 the decoder was trained on.  The row then holds c = x clamped into the cube
// and no extrapolated decoder value is ever trusted: the rendered surface lies inside the cube (the rays are clipped to it), the cube is
// convex and c is its nearest point to x, so for every surface point s:  |x - s|^2 >= |x - c|^2 + |c - s|^2 >= cd^2 + max(f(c), 0)^2
// (cone_value below).  Inside the cube cd is exactly 0 and the value is the decoder's.
__device__ __forceinline__ float cone_write_row(float* __restrict__ row, const float* __restrict__ latn, int L, const TraceRay& r, float lam, float bound) {
    for (int c = 0; c < L; ++c) row[c] = latn[c];
    const float x = r.ox + lam * r.dx, y = r.oy + lam * r.dy, z = r.oz + lam * r.dz;
    const float cx = fminf(fmaxf(x, -bound), bound), cy = fminf(fmaxf(y, -bound), bound), cz = fminf(fmaxf(z, -bound), bound);
    row[L] = cx; row[L + 1] = cy; row[L + 2] = cz;
    const float ex = x - cx, ey = y - cy, ez = z - cz;
    return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
}
__device__ __forceinline__ float cone_value(float v, float cd) {
    if (!(cd > 0.f)) return v;
    const float vp = fmaxf(v, 0.f);
    return sqrtf(__fadd_rn(__fmul_rn(cd, cd), __fmul_rn(vp, vp)));
}
__device__ __forceinline__ float cone_clamp_dist(const TraceRay& r, float lam, float bound) {
    const float x = r.ox + lam * r.dx, y = r.oy + lam * r.dy, z = r.oz + lam * r.dz;
    const float ex = x - fminf(fmaxf(x, -bound), bound), ey = y - fminf(fmaxf(y, -bound), bound), ez = z - fminf(fmaxf(z, -bound), bound);
    return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
}

// slab test of a ray against the cube [-bound, bound]^3 the SDF is defined on: entry (>= near) and exit parameters
__device__ __forceinline__ bool trace_slab(const TraceRay& r, float bound, float near, float& l0, float& l1) {
    l0 = near; l1 = FLT_MAX;
    const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
    bool active = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (fabsf(d[a]) < 1e-12f) { active = active && (fabsf(o[a]) <= bound); continue; }
        float ta = (-bound - o[a]) / d[a], tb = (bound - o[a]) / d[a];
        if (ta > tb) { const float t = ta; ta = tb; tb = t; }
        l0 = fmaxf(l0, ta); l1 = fminf(l1, tb);
    }
    return active && (l0 < l1);
}

// every pixel of every crop: slab test against the cube; rays that hit it enter the active list.  cone (optional, sdfr_trace_cone): per
// cone_block x cone_block pixel block the parameter its cone march stopped at (the block's rays start there) or -1 (no ray of the block can hit)
__global__ __launch_bounds__(256) void sdfr_trace_setup_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                              const float* __restrict__ latn, int L, const TraceDims D, float bound, float near,
                                                              int32_t* __restrict__ counters, int32_t* __restrict__ pix,
                                                              float4* __restrict__ lam, float* __restrict__ far, float* __restrict__ inputs,
                                                              const float* __restrict__ cone, int cone_block, float* __restrict__ hit_lam,
                                                              float* __restrict__ hit_sdf) {
    const int b = blockIdx.y;
    const int P_ = D.PS;
    int W, H;
    trace_dims(D, b, W, H);
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool active = false;
    TraceRay r = {};
    float l0 = 0.f, l1 = 0.f;
    if (p < W * H) {
        const int x = p % W, y = p / W;
        r = trace_ray(pose + (int64_t)b * 16, Kinv + (int64_t)b * 9, (float)x, (float)y);
        active = trace_slab(r, bound, near, l0, l1);
        if (cone && active) {
            const int nbx = (W + cone_block - 1) / cone_block;
            const float c = cone[(int64_t)b * D.ncap + (y / cone_block) * nbx + x / cone_block];
            if (c < 0.f) active = false;
            else { l0 = fmaxf(l0, c); active = l0 < l1; }
        }
    }
    if (p < P_) {
        far[(int64_t)b * P_ + p] = active ? l1 : 0.f;
        if (hit_lam) { hit_lam[(int64_t)b * P_ + p] = 0.f; hit_sdf[(int64_t)b * P_ + p] = 0.f; }     // (the march records hits here: no separate fills)
    }
    const int slot = trace_append(active, counters);
    if (active) {
        pix[slot] = b * P_ + p;
        lam[slot] = make_float4(l0, 0.f, 1.f, 0.f);
        trace_write_row(inputs + (int64_t)slot * (L + 3), latn + (int64_t)b * L, L, r, l0);
    }
}

// ---- cone marching (optional first phase): ONE ray through the centre of a block x block pixel tile stands for all its pixels ----------------
// All pixel rays share the origin and the parametrisation (lam = camera depth), so the block's rays at parameter lam lie within lam * delta of
// the centre ray's point (delta = max over the block's corner pixels of |d_corner - d_centre|).  v = decoder(centre point): free = v - lam delta
// > eps -> no surface within the cone's cross-section there, advance to lam + free / (|d_c| + delta); free <= eps -> the block's rays start their
// own march at lam; past the far side of the cube for ALL the block's rays -> culled (oracle/sdf_oracle.py::cone_march).
// cone state float4: (lam, delta, far of the block, |d_c|); list entry: crop * blocks + block
__device__ __forceinline__ TraceRay cone_centre_ray(const float* __restrict__ P, const float* __restrict__ Ki, int k, int nbx, int BL, int W, int H,
                                                   int& x0, int& y0, int& x1, int& y1) {
    x0 = (k % nbx) * BL; y0 = (k / nbx) * BL;
    x1 = min(x0 + BL - 1, W - 1); y1 = min(y0 + BL - 1, H - 1);
    return trace_ray(P, Ki, 0.5f * (float)(x0 + x1), 0.5f * (float)(y0 + y1));
}

// Speculative cone passes (r04): a pass evaluates K samples of a cone, p_0 = lam, p_j = p_{j-1} + sigma q^j a_prev (a_prev = the cone's previous
// advance, q = the ratio of its last two advances clamped to [0.5, 1.5]); sample j counts only inside the range its predecessor proved free.
// Rows of cone slot s: s * K + j.  The rotating counters hold ROW counts (K per listed cone).
__device__ __forceinline__ void cone_positions(float lam, float aprev, float q, int K, float sigma, float* __restrict__ p) {
    float pj = lam, qp = q;
    for (int j = 0; j < K; ++j) {
        if (j > 0) { pj = __fadd_rn(pj, __fmul_rn(__fmul_rn(sigma, qp), aprev)); qp = __fmul_rn(qp, q); }
        p[j] = pj;
    }
}
#define CONE_KMAX 8
__device__ __forceinline__ int cone_append(bool keep, int32_t* __restrict__ counter, int K) {
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(counter, __popcll(bal) * K);
    base = __shfl(base, 0, 64);
    return base / K + __popcll(bal & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(256) void sdfr_trace_cone_setup_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                                   const float* __restrict__ latn, int L, const TraceDims D, int BL, float bound,
                                                                   float near, int K, float sigma, int32_t* __restrict__ counters,
                                                                   int32_t* __restrict__ ids, float4* __restrict__ st, float2* __restrict__ aux,
                                                                   float* __restrict__ cone, float* __restrict__ inputs) {
    const int b = blockIdx.y;
    int W, H;
    trace_dims(D, b, W, H);
    const int nbx = (W + BL - 1) / BL, nby = (H + BL - 1) / BL, nblk = nbx * nby;
    const int k = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    TraceRay rc = {};
    float near_b = FLT_MAX, far_b = 0.f, delta = 0.f;
    if (k < nblk) {
        const float* Pm = pose + (int64_t)b * 16;
        const float* Ki = Kinv + (int64_t)b * 9;
        int x0, y0, x1, y1;
        rc = cone_centre_ray(Pm, Ki, k, nbx, BL, W, H, x0, y0, x1, y1);
        const int cxs[4] = {x0, x1, x0, x1}, cys[4] = {y0, y0, y1, y1};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const TraceRay rk = trace_ray(Pm, Ki, (float)cxs[c], (float)cys[c]);
            const float ex = rk.dx - rc.dx, ey = rk.dy - rc.dy, ez = rk.dz - rc.dz;
            delta = fmaxf(delta, sqrtf(ex * ex + ey * ey + ez * ez));
        }
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const TraceRay rk = trace_ray(Pm, Ki, (float)x, (float)y);
                float l0, l1;
                if (trace_slab(rk, bound, near, l0, l1)) { near_b = fminf(near_b, l0); far_b = fmaxf(far_b, l1); keep = true; }
            }
        cone[(int64_t)b * D.ncap + k] = -1.f;                   // until the march says where the block's rays start
    }
    const int slot = cone_append(keep, counters, K);
    if (keep) {
        const float a0 = __fmul_rn(0.1f, __fsub_rn(far_b, near_b));      // first guess of an advance: a tenth of the block's parameter range
        ids[slot] = b * D.ncap + k;
        st[slot] = make_float4(near_b, delta, far_b, sqrtf(rc.dx * rc.dx + rc.dy * rc.dy + rc.dz * rc.dz));
        aux[slot] = make_float2(a0, 1.f);
        float p[CONE_KMAX];
        cone_positions(near_b, a0, 1.f, K, sigma, p);
        for (int j = 0; j < K; ++j)
            (void)cone_write_row(inputs + ((int64_t)slot * K + j) * (L + 3), latn + (int64_t)b * L, L, rc, p[j], bound);
    }
}

__global__ __launch_bounds__(256) void sdfr_trace_cone_step_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                                  const float* __restrict__ latn, int L, const TraceDims D, int BL, float eps,
                                                                  float bound, int K, float sigma, const float* __restrict__ sdf,
                                                                  const int32_t* __restrict__ n_cur, int32_t* __restrict__ n_next,
                                                                  int32_t* __restrict__ n_zero, const int32_t* __restrict__ ids_in,
                                                                  const float4* __restrict__ st_in, const float2* __restrict__ aux_in,
                                                                  int32_t* __restrict__ ids_out, float4* __restrict__ st_out,
                                                                  float2* __restrict__ aux_out, float* __restrict__ inputs,
                                                                  float* __restrict__ cone, int last, unsigned long long* __restrict__ evals) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int nrows = *n_cur;
    const int n = nrows / K;
    if (s == 0) *n_zero = 0;
    if (s == 0 && evals) atomicAdd(evals, (unsigned long long)nrows);
    if (blockIdx.x * 256 >= n) return;
    bool keep = false;
    int id = 0;
    float4 st = make_float4(0.f, 0.f, 0.f, 1.f);
    float2 ax = make_float2(0.f, 1.f);
    TraceRay rc = {};
    int b = 0;
    if (s < n) {
        id = ids_in[s];
        st = st_in[s];
        ax = aux_in[s];
        b = id / D.ncap;
        int W, H;
        trace_dims(D, b, W, H);
        const int nbx = (W + BL - 1) / BL;
        int x0, y0, x1, y1;
        rc = cone_centre_ray(pose + (int64_t)b * 16, Kinv + (int64_t)b * 9, id - b * D.ncap, nbx, BL, W, H, x0, y0, x1, y1);
        float p[CONE_KMAX];
        cone_positions(st.x, ax.x, ax.y, K, sigma, p);
        const float den = __fadd_rn(st.w, st.y);
        // walk the accepted prefix (oracle/sdf_oracle.py::cone_march)
        float lam_new = st.x, a_last = ax.x, a_before = ax.x, a_prevj = 0.f;
        bool done = false, alive = true;
        for (int j = 0; j < K && !done; ++j) {
            if (j > 0) {
                const float gap = __fsub_rn(p[j], p[j - 1]);
                if (!((p[j] > p[j - 1]) && (gap <= a_prevj))) break;       // the prediction left the range its predecessor proved free
            }
            const float v = cone_value(sdf[(int64_t)s * K + j], cone_clamp_dist(rc, p[j], bound));   // (the row held the point clamped into the cube)
            const float free_ = __fsub_rn(v, __fmul_rn(p[j], st.y));
            if (!(free_ > eps)) { cone[id] = p[j]; alive = false; done = true; break; }   // touches the tolerance band (or NaN): the rays take over here
            const float a = free_ / den;
            const float adv = __fadd_rn(p[j], a);
            if (!(adv < st.z)) { cone[id] = -1.f; alive = false; done = true; break; }     // past the cube for every ray of the block: culled
            a_before = a_last; a_last = a; lam_new = adv; a_prevj = a;
        }
        if (alive) {
            const float qn = (a_before > 0.f) ? fminf(fmaxf(a_last / a_before, 0.5f), 1.5f) : 1.f;
            if (last) cone[id] = lam_new;                          // out of cone passes: the rays start where the cone got to
            else { keep = true; st.x = lam_new; ax = make_float2(a_last, qn); }
        }
    }
    const int slot = cone_append(keep, n_next, K);
    if (keep) {
        ids_out[slot] = id;
        st_out[slot] = st;
        aux_out[slot] = ax;
        float p[CONE_KMAX];
        cone_positions(st.x, ax.x, ax.y, K, sigma, p);
        for (int j = 0; j < K; ++j)
            (void)cone_write_row(inputs + ((int64_t)slot * K + j) * (L + 3), latn + (int64_t)b * L, L, rc, p[j], bound);
    }
}

// one march step of every active ray (count on the device): advance by the decoder's value, retire hits and exits, compact the survivors
__global__ __launch_bounds__(256) void sdfr_trace_step_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                             const float* __restrict__ latn, int L, const TraceDims D, float eps,
                                                             const float* __restrict__ sdf, const int32_t* __restrict__ n_cur,
                                                             int32_t* __restrict__ n_next, int32_t* __restrict__ n_zero,
                                                             const int32_t* __restrict__ pix_in, const float4* __restrict__ lam_in,
                                                             int32_t* __restrict__ pix_out, float4* __restrict__ lam_out,
                                                             const float* __restrict__ far, float* __restrict__ inputs,
                                                             float* __restrict__ hit_lam, float* __restrict__ hit_sdf, int min_count,
                                                             unsigned long long* __restrict__ evals, float qmax) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int n = *n_cur;
    if (s == 0) *n_zero = 0;                     // the counter of the step after next (three counters rotate)
    if (n < min_count) return;                   // fewer rays than the tail threshold: the persistent tail kernel marches them to the end
    if (s == 0 && evals) atomicAdd(evals, (unsigned long long)n);
    if (blockIdx.x * 256 >= n) return;
    bool keep = false;
    int gp = 0;
    float4 st = make_float4(0.f, 0.f, 0.f, 1.f);
    TraceRay r = {};
    if (s < n) {
        gp = pix_in[s];
        const int P_ = D.PS, b = gp / P_, p = gp - b * P_;
        int W, H;
        trace_dims(D, b, W, H);
        r = trace_ray(pose + (int64_t)b * 16, Kinv + (int64_t)b * 9, (float)(p % W), (float)(p / W));
        const float v = sdf[s];
        st = lam_in[s];
        const int what = trace_advance(st, v, sqrtf(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz), eps, far[gp], qmax);
        if (what == 1) {                         // on the surface: retire as a hit
            hit_lam[gp] = st.x;
            hit_sdf[gp] = v;
        }
        keep = what == 0;                        // (-1: past the cube or NaN: a miss)
    }
    const int slot = trace_append(keep, n_next);
    if (keep) {
        const int b = gp / D.PS;
        pix_out[slot] = gp;
        lam_out[slot] = st;
        trace_write_row(inputs + (int64_t)slot * (L + 3), latn + (int64_t)b * L, L, r, st.x);
    }
}

__global__ void sdfr_trace_leftover_kernel(const int32_t* __restrict__ n_cur, int32_t* __restrict__ unresolved) {
    if (threadIdx.x == 0 && *n_cur > 0) atomicAdd(unresolved, *n_cur);
}

// ---- after the march: hit list, images, gradients ------------------------------------------------------------------------------------------
// hit pixels -> compact list of decoder rows [latent, x0 = o + lam0 d] (ballot append; the order is irrelevant: everything downstream is
// addressed per pixel through hit_slot) for the exact-f32 value + Jacobian pass (sdfr_mlp_jacobian with idx = identity, cnt = n_hits)
__global__ __launch_bounds__(256) void sdfr_trace_hits_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                                             const float* __restrict__ latn, int L, const TraceDims D,
                                                             const float* __restrict__ hit_lam, int32_t* __restrict__ n_hits,
                                                             int32_t* __restrict__ hit_slot, int32_t* __restrict__ idx, float* __restrict__ rows) {
    const int b = blockIdx.y, P_ = D.PS;
    int W, H;
    trace_dims(D, b, W, H);
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    float lam = 0.f;
    TraceRay r = {};
    if (p < P_) {
        lam = hit_lam[(int64_t)b * P_ + p];
        hit = lam > 0.f && p < W * H;
        if (hit) r = trace_ray(pose + (int64_t)b * 16, Kinv + (int64_t)b * 9, (float)(p % W), (float)(p / W));
    }
    const int slot = trace_append(hit, n_hits);
    if (p < P_) hit_slot[(int64_t)b * P_ + p] = hit ? slot : -1;
    if (hit) {
        idx[slot] = slot;
        trace_write_row(rows + (int64_t)slot * (L + 3), latn + (int64_t)b * L, L, r, lam);
    }
}

struct TraceHit { float lam_s, c, nx, ny, nz; };      // polished ray parameter, 1 / (grad f . d) (0: grazing, no implicit term), unit normal

// One Newton step along the ray from the marched point with the exact-f32 decoder value f0 and input gradient (gx = d f / d x) there:
// lam_s = lam0 - f0 / (gx . d) -- only for rays that meet the surface at more than ~6 degrees (|gx . d| > 0.1 |gx| |d|); along a grazing ray
// the first-order step is long and leaves the linear region of the decoder: those hits keep the marched point (|f| < eps).
__device__ __forceinline__ TraceHit trace_polish(const TraceRay& r, float lam0, float f0, const float* __restrict__ Jrow, int L) {
    const float gx = Jrow[L], gy = Jrow[L + 1], gz = Jrow[L + 2];
    const float gd = gx * r.dx + gy * r.dy + gz * r.dz;
    const float gn = sqrtf(gx * gx + gy * gy + gz * gz), dn = sqrtf(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz);
    const bool ok = fabsf(gd) > 0.1f * gn * dn;
    TraceHit h;
    h.lam_s = ok ? lam0 - f0 / gd : lam0;
    h.c = ok ? 1.f / gd : 0.f;
    const float inv = 1.f / fmaxf(gn, 1e-12f);                    // F.normalize
    h.nx = gx * inv; h.ny = gy * inv; h.nz = gz * inv;
    return h;
}

// images of the hits: depth = lam_s r_z, NOCS colour = (x_s (-1,1,1) + 1) / 2 (projection.py:53-55, rasterer.py:113-114), normals = (R n + 1) / 2, mask = 1;
// zero elsewhere.  Layouts as the splat renderer's: color [B][3][H][W], mask / depth [B][1][H][W], normals [B][3][H][W].
__global__ __launch_bounds__(256) void sdfr_trace_composite_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv, int L, const TraceDims D,
                                                                  const float* __restrict__ hit_lam, const int32_t* __restrict__ hit_slot,
                                                                  const float* __restrict__ J, const float* __restrict__ f0,
                                                                  float* __restrict__ color, float* __restrict__ mask, float* __restrict__ depth,
                                                                  float* __restrict__ normals, float* __restrict__ lam_s) {
    const int b = blockIdx.y, P_ = D.PS;
    int W, H;
    trace_dims(D, b, W, H);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P_) return;
    const int64_t gp = (int64_t)b * P_ + p;
    const int slot = hit_slot[gp];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, dep = 0.f, m = 0.f, ls = 0.f;
    if (slot >= 0) {
        const float* Pm = pose + (int64_t)b * 16;
        const float* Ki = Kinv + (int64_t)b * 9;
        const float x = (float)(p % W), y = (float)(p / W);
        const TraceRay r = trace_ray(Pm, Ki, x, y);
        const TraceHit h = trace_polish(r, hit_lam[gp], f0[slot], J + (int64_t)slot * (L + 3), L);
        const float rz = fmaf(Ki[7], y, Ki[6] * x) + Ki[8];
        const float xs = r.ox + h.lam_s * r.dx, ys = r.oy + h.lam_s * r.dy, zs = r.oz + h.lam_s * r.dz;
        c0 = (-xs + 1.f) / 2.f; c1 = (ys + 1.f) / 2.f; c2 = (zs + 1.f) / 2.f;
        n0 = (Pm[0] * h.nx + Pm[1] * h.ny + Pm[2] * h.nz + 1.f) / 2.f;
        n1 = (Pm[4] * h.nx + Pm[5] * h.ny + Pm[6] * h.nz + 1.f) / 2.f;
        n2 = (Pm[8] * h.nx + Pm[9] * h.ny + Pm[10] * h.nz + 1.f) / 2.f;
        dep = h.lam_s * rz; m = 1.f; ls = h.lam_s;
    }
    float* cb = color + (int64_t)b * 3 * P_;
    float* nb = normals + (int64_t)b * 3 * P_;
    cb[p] = c0; cb[P_ + p] = c1; cb[2 * P_ + p] = c2;
    nb[p] = n0; nb[P_ + p] = n1; nb[2 * P_ + p] = n2;
    mask[gp] = m; depth[gp] = dep;
    if (lam_s) lam_s[gp] = ls;
}

// Backward at a fixed hit set.  The hit depth is an implicit function of pose and latent, f(o(θ) + λ d(θ), z(θ)) = 0:
//     λ(θ) = λ_s - c [ gx . (o(θ) + λ_s d(θ) - x_s) + gz . (z(θ) - z_s) ],   c = 1 / (gx . d)   (c = 0 for grazing hits: λ constant)
//     x(θ) = o(θ) + λ(θ) d(θ),   o = -R^T t,   d = R^T r,   n_cam = R n (n constant, as the splat path's normals: grid.py:57-58)
// Given the image gradients, every hit pixel contributes to d L / d R (9), d L / d t (3) and d L / d z (L); the contributions of a crop are summed
// in a FIXED order (block tree here, block partials in order in the second kernel): deterministic, batch independent.
#define TRB_THREADS 256
#define TRB_MAXL 8
__global__ __launch_bounds__(TRB_THREADS) void sdfr_trace_backward_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv, int L,
                                                                         const TraceDims D, const float* __restrict__ hit_lam,
                                                                         const int32_t* __restrict__ hit_slot, const float* __restrict__ J,
                                                                         const float* __restrict__ f0, const float* __restrict__ g_color,
                                                                         const float* __restrict__ g_depth, const float* __restrict__ g_normals,
                                                                         const float* __restrict__ g_xyzf, const int32_t* __restrict__ pt_slot,
                                                                         int ecap, int surfel, float* __restrict__ partial) {
    constexpr int NV = 12 + TRB_MAXL;
    const int b = blockIdx.y, P_ = D.PS, tid = threadIdx.x;
    int W, H;
    trace_dims(D, b, W, H);
    (void)H;
    const int p = blockIdx.x * TRB_THREADS + tid;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    const int64_t gp = (int64_t)b * P_ + p;
    const int slot = p < P_ ? hit_slot[gp] : -1;
    if (slot >= 0) {
        const float* Pm = pose + (int64_t)b * 16;
        const float* Ki = Kinv + (int64_t)b * 9;
        const float x = (float)(p % W), y = (float)(p / W);
        const TraceRay r = trace_ray(Pm, Ki, x, y);
        const float* Jr = J + (int64_t)slot * (L + 3);
        const TraceHit h = trace_polish(r, hit_lam[gp], f0[slot], Jr, L);
        const float rx = fmaf(Ki[1], y, Ki[0] * x) + Ki[2], ry = fmaf(Ki[4], y, Ki[3] * x) + Ki[5], rz = fmaf(Ki[7], y, Ki[6] * x) + Ki[8];
        // upstream: d L / d x (through the NOCS colour), d L / d λ, d L / d n_cam
        float gxs[3] = {0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f};
        if (g_color) { const float* g = g_color + (int64_t)b * 3 * P_; gxs[0] = -g[p] / 2.f; gxs[1] = g[P_ + p] / 2.f; gxs[2] = g[2 * P_ + p] / 2.f; }
        if (g_normals) { const float* g = g_normals + (int64_t)b * 3 * P_; gn[0] = g[p] / 2.f; gn[1] = g[P_ + p] / 2.f; gn[2] = g[2 * P_ + p] / 2.f; }
        // gradient arriving through the hit's camera-frame point (points['xyzf'] of the refinement loop: sdfr_trace_points)
        float ge[3] = {0.f, 0.f, 0.f};
        if (g_xyzf) {
            const int ps = pt_slot[gp];
            if (ps >= 0 && ps < ecap) { const float* g = g_xyzf + ((int64_t)b * ecap + ps) * 3; ge[0] = g[0]; ge[1] = g[1]; ge[2] = g[2]; }
        }
        const float nh[3] = {h.nx, h.ny, h.nz};
        if (!surfel) {
            // image-space derivative at the fixed pixel: p_cam = lam r with the pixel ray r fixed, so the point only moves along its ray
            float gl = gxs[0] * r.dx + gxs[1] * r.dy + gxs[2] * r.dz + (ge[0] * rx + ge[1] * ry + ge[2] * rz);
            if (g_depth) gl += g_depth[gp] * rz;
            // d λ = -c [ gx . (d o + λ_s d d) + gz . d z ]  ->  adjoint on w = o + λ_s d:  g_w = g_x - c g_λ gx
            const float k = h.c * gl;
            const float gw[3] = {gxs[0] - k * Jr[L], gxs[1] - k * Jr[L + 1], gxs[2] - k * Jr[L + 2]};
            const float go[3] = {gw[0], gw[1], gw[2]};
            const float gd[3] = {h.lam_s * gw[0], h.lam_s * gw[1], h.lam_s * gw[2]};
            const float t[3] = {Pm[3], Pm[7], Pm[11]}, rr[3] = {rx, ry, rz};
            // o_j = -sum_i R_ij t_i,  d_j = sum_i R_ij r_i,  n_cam_i = sum_j R_ij n_j
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) v[i * 3 + j] = -t[i] * go[j] + rr[i] * gd[j] + gn[i] * nh[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[9 + i] = -(Pm[i * 4] * go[0] + Pm[i * 4 + 1] * go[1] + Pm[i * 4 + 2] * go[2]);
#pragma unroll
            for (int c = 0; c < TRB_MAXL; ++c)
                if (c < L) v[12 + c] = -k * Jr[c];
        } else {
            // surfel semantics -- the autograd semantics of the reference's surface points (grid.py:61 p = x - sdf n_hat with n_hat constant,
            // projection.py:53-58 colour = the point's own object coordinates, p_cam = R p + t): the hit point x_s is a MATERIAL point that
            // moves rigidly with the pose and along its normal with the latent, d x_s = -n_hat (gz . dz) / |gx|; its colour does not depend
            // on the pose at all.  depth = (R x_s + t)_z.
            if (g_depth) ge[2] += g_depth[gp];
            const float xs[3] = {r.ox + h.lam_s * r.dx, r.oy + h.lam_s * r.dy, r.oz + h.lam_s * r.dz};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) v[i * 3 + j] = ge[i] * xs[j] + gn[i] * nh[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[9 + i] = ge[i];
            // object-space gradient on x_s: R^T ge + the colour's
            const float gox = Pm[0] * ge[0] + Pm[4] * ge[1] + Pm[8] * ge[2] + gxs[0];
            const float goy = Pm[1] * ge[0] + Pm[5] * ge[1] + Pm[9] * ge[2] + gxs[1];
            const float goz = Pm[2] * ge[0] + Pm[6] * ge[1] + Pm[10] * ge[2] + gxs[2];
            const float gnorm = sqrtf(Jr[L] * Jr[L] + Jr[L + 1] * Jr[L + 1] + Jr[L + 2] * Jr[L + 2]);
            const float k = (gox * nh[0] + goy * nh[1] + goz * nh[2]) / fmaxf(gnorm, 1e-12f);
#pragma unroll
            for (int c = 0; c < TRB_MAXL; ++c)
                if (c < L) v[12 + c] = -k * Jr[c];
        }
    }
    __shared__ float red[NV][TRB_THREADS];
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i][tid] = v[i];
    __syncthreads();
    for (int o = TRB_THREADS / 2; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int i = 0; i < NV; ++i) red[i][tid] += red[i][tid + o];
        }
        __syncthreads();
    }
    if (tid < NV) partial[((int64_t)b * gridDim.x + blockIdx.x) * NV + tid] = red[tid][0];
}

// second stage: the block partials of a crop, summed in a fixed order (thread t takes partials t, t + 256, ...; then a tree over the threads)
__global__ __launch_bounds__(256) void sdfr_trace_backward_sum_kernel(const float* __restrict__ partial, int nblk, int L, float* __restrict__ g_pose,
                                                                     float* __restrict__ g_latn) {
    constexpr int NV = 12 + TRB_MAXL;
    const int b = blockIdx.x, tid = threadIdx.x;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    for (int k = tid; k < nblk; k += 256) {
        const float* p = partial + ((int64_t)b * nblk + k) * NV;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += p[i];
    }
    __shared__ float red[NV][256];
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i][tid] = v[i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int i = 0; i < NV; ++i) red[i][tid] += red[i][tid + o];
        }
        __syncthreads();
    }
    if (tid < NV) {
        const float s = red[tid][0];
        if (tid < 9) g_pose[(int64_t)b * 16 + (tid / 3) * 4 + (tid % 3)] = s;
        else if (tid < 12) g_pose[(int64_t)b * 16 + (tid - 9) * 4 + 3] = s;
        else if (tid - 12 < L) g_latn[(int64_t)b * L + (tid - 12)] = s;
    }
    if (tid >= 32 && tid < 36) g_pose[(int64_t)b * 16 + 12 + (tid - 32)] = 0.f;
}

// ---- extents of a call: dense (W, H) or ragged (sdfr_extents: per-crop sizes on the device) ---------------------------------------------------
static inline TraceDims dense_dims(int W, int H, int cone_block) {
    TraceDims D = {nullptr, W, H, W * H, cone_block > 0 ? sdfr_cdiv(W, cone_block) * sdfr_cdiv(H, cone_block) : 1};
    return D;
}
static inline int ragged_dims(TraceDims& D, const sdfr_extents* e, const char* who) {
    SDFR_REQUIRE(e && e->wh && e->pix_stride > 0 && e->cone_cap > 0, "%s: ragged extents need wh, pix_stride and cone_cap", who);
    D.wh = e->wh; D.W = 0; D.H = 0; D.PS = e->pix_stride; D.ncap = e->cone_cap;
    return SDFR_OK;
}

extern "C" int sdfr_trace_setup2(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound, float near,
                                 int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block,
                                 float* hit_lam, float* hit_sdf, void* stream);
extern "C" int sdfr_trace_setup(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound, float near,
                                int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block,
                                void* stream) {
    return sdfr_trace_setup2(pose, Kinv, latn, L, B, W, H, bound, near, counters, pix, lam, far, inputs, cone, cone_block, nullptr, nullptr, stream);
}

// ... which also zero-fills the march's per-pixel hit records (hit_lam / hit_sdf float[B*W*H], both or neither): two launches fewer per render
static int trace_setup_impl(const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D, float bound, float near,
                            int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block,
                            float* hit_lam, float* hit_sdf, void* stream);
extern "C" int sdfr_trace_setup2(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound, float near,
                                 int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block,
                                 float* hit_lam, float* hit_sdf, void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0, "sdfr_trace_setup: bad size");
    return trace_setup_impl(pose, Kinv, latn, L, B, dense_dims(W, H, cone_block), bound, near, counters, pix, lam, far, inputs, cone, cone_block, hit_lam,
                            hit_sdf, stream);
}
extern "C" int sdfr_trace_setup_r(const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext, float bound, float near,
                                  int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block,
                                  float* hit_lam, float* hit_sdf, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_setup_r");
    if (rc) return rc;
    return trace_setup_impl(pose, Kinv, latn, L, B, D, bound, near, counters, pix, lam, far, inputs, cone, cone_block, hit_lam, hit_sdf, stream);
}
static int trace_setup_impl(const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D, float bound, float near,
                            int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block,
                            float* hit_lam, float* hit_sdf, void* stream) {
    SDFR_REQUIRE((hit_lam == nullptr) == (hit_sdf == nullptr), "sdfr_trace_setup2: hit_lam and hit_sdf go together");
    SDFR_REQUIRE(pose && Kinv && latn && counters && pix && lam && far && inputs, "sdfr_trace_setup: NULL argument");
    SDFR_REQUIRE(L >= 0 && B > 0 && bound > 0.f, "sdfr_trace_setup: bad size");
    SDFR_REQUIRE(!cone || cone_block >= 2, "sdfr_trace_setup: cone starts need their block size (>= 2), got %d", cone_block);
    hipStream_t s = (hipStream_t)stream;
    SDFR_HIP_CHECK(sdfr_zero_async(counters, SDFR_TRACE_COUNTERS * sizeof(int32_t), s));
    hipLaunchKernelGGL(sdfr_trace_setup_kernel, dim3(sdfr_cdiv((int64_t)D.PS, 256), B), dim3(256), 0, s, pose, Kinv, latn, L, D, bound, near,
                       counters, pix, reinterpret_cast<float4*>(lam), far, inputs, cone, cone_block, hit_lam, hit_sdf);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// Cone marching ahead of the per-ray march: cone float[B][ceil(W/block)*ceil(H/block)] receives per pixel block the parameter its rays start
// from, or -1 (no ray of the block can hit).  counters: device int32[8] (zeroed here; [0..2] rotating counts, [4..5] one uint64: decoder
// evaluations), ids0/st0, ids1/st1: ping-pong cone lists (int32[n] / float[n][4], n = B * blocks), inputs float[n][L+3], sdf float[n].
static int trace_cone_impl(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D,
                           float bound, float near, float eps, int block, int cone_steps, int spec_k, float sigma, int half,
                           int32_t* counters, int32_t* ids0, float* st0, float* aux0, int32_t* ids1, float* st1, float* aux1, float* inputs,
                           float* sdf, float* cone, void* stream);
extern "C" int sdfr_trace_cone(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H,
                               float bound, float near, float eps, int block, int cone_steps, int spec_k, float sigma, int half,
                               int32_t* counters, int32_t* ids0, float* st0, float* aux0, int32_t* ids1, float* st1, float* aux1, float* inputs,
                               float* sdf, float* cone, void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0 && block >= 2, "sdfr_trace_cone: bad size");
    return trace_cone_impl(d, pose, Kinv, latn, L, B, dense_dims(W, H, block), bound, near, eps, block, cone_steps, spec_k, sigma, half, counters, ids0, st0,
                           aux0, ids1, st1, aux1, inputs, sdf, cone, stream);
}
extern "C" int sdfr_trace_cone_r(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext,
                                 float bound, float near, float eps, int block, int cone_steps, int spec_k, float sigma, int half,
                                 int32_t* counters, int32_t* ids0, float* st0, float* aux0, int32_t* ids1, float* st1, float* aux1, float* inputs,
                                 float* sdf, float* cone, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_cone_r");
    if (rc) return rc;
    return trace_cone_impl(d, pose, Kinv, latn, L, B, D, bound, near, eps, block, cone_steps, spec_k, sigma, half, counters, ids0, st0, aux0, ids1, st1,
                           aux1, inputs, sdf, cone, stream);
}
static int trace_cone_impl(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D,
                           float bound, float near, float eps, int block, int cone_steps, int spec_k, float sigma, int half,
                           int32_t* counters, int32_t* ids0, float* st0, float* aux0, int32_t* ids1, float* st1, float* aux1, float* inputs,
                           float* sdf, float* cone, void* stream) {
    SDFR_REQUIRE(d && pose && Kinv && latn && counters && ids0 && st0 && aux0 && ids1 && st1 && aux1 && inputs && sdf && cone,
                 "sdfr_trace_cone: NULL argument");
    SDFR_REQUIRE(L >= 0 && B > 0 && bound > 0.f && block >= 2 && cone_steps >= 1, "sdfr_trace_cone: bad size");
    SDFR_REQUIRE(spec_k >= 1 && spec_k <= CONE_KMAX, "sdfr_trace_cone: spec_k = %d (1 ... %d samples per cone and pass)", spec_k, CONE_KMAX);
    SDFR_REQUIRE(d->n_inputs == L + 3, "sdfr_trace_cone: decoder with L + 3 = %d inputs expected, it has %d", L + 3, d->n_inputs);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = D.ncap;
    const int64_t n_max = (int64_t)B * nblk;
    SDFR_REQUIRE(n_max * spec_k < (int64_t)1 << 31, "sdfr_trace_cone: too many cone rows");
    SDFR_HIP_CHECK(sdfr_zero_async(counters, SDFR_TRACE_COUNTERS * sizeof(int32_t), s));
    hipLaunchKernelGGL(sdfr_trace_cone_setup_kernel, dim3(sdfr_cdiv(nblk, 256), B), dim3(256), 0, s, pose, Kinv, latn, L, D, block, bound, near,
                       spec_k, sigma, counters, ids0, reinterpret_cast<float4*>(st0), reinterpret_cast<float2*>(aux0), cone, inputs);
    unsigned long long* evals = reinterpret_cast<unsigned long long*>(counters + 4);
    for (int step = 0; step < cone_steps; ++step) {
        const int rc = sdfr_mlp_forward_counted(d, inputs, n_max * spec_k, counters + step % 3, sdf, half, stream);
        if (rc != SDFR_OK) return rc;
        const int a = step & 1;
        hipLaunchKernelGGL(sdfr_trace_cone_step_kernel, dim3(sdfr_cdiv(n_max, 256)), dim3(256), 0, s, pose, Kinv, latn, L, D, block, eps, bound,
                           spec_k, sigma, sdf, counters + step % 3, counters + (step + 1) % 3, counters + (step + 2) % 3, a ? ids1 : ids0,
                           reinterpret_cast<const float4*>(a ? st1 : st0), reinterpret_cast<const float2*>(a ? aux1 : aux0), a ? ids0 : ids1,
                           reinterpret_cast<float4*>(a ? st0 : st1), reinterpret_cast<float2*>(a ? aux0 : aux1), inputs, cone,
                           step == cone_steps - 1 ? 1 : 0, evals);
    }
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_trace_step(const float* pose, const float* Kinv, const float* latn, int L, int W, int H, float eps,
                               const float* sdf, int32_t* counters, int step, int64_t n_max, const int32_t* pix_in, const float* lam_in,
                               int32_t* pix_out, float* lam_out, const float* far, float* inputs, float* hit_lam, float* hit_sdf, void* stream) {
    SDFR_REQUIRE(pose && Kinv && latn && sdf && counters && pix_in && lam_in && pix_out && lam_out && far && inputs && hit_lam && hit_sdf,
                 "sdfr_trace_step: NULL argument");
    SDFR_REQUIRE(step >= 0 && n_max >= 0, "sdfr_trace_step: bad size");
    if (n_max == 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_trace_step_kernel, dim3(sdfr_cdiv(n_max, 256)), dim3(256), 0, (hipStream_t)stream, pose, Kinv, latn, L,
                       dense_dims(W, H, 0), eps, sdf, counters + step % 3, counters + (step + 1) % 3, counters + (step + 2) % 3, pix_in,
                       reinterpret_cast<const float4*>(lam_in), pix_out, reinterpret_cast<float4*>(lam_out), far, inputs, hit_lam, hit_sdf, 0,
                       (unsigned long long*)nullptr, 1.f);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- the whole march in one call: no host synchronisation -------------------------------------------------------------------------------
// counters (device int32[SDFR_TRACE_COUNTERS], zeroed by sdfr_trace_setup): [0..2] rotating active counts, [3] rays left unresolved when the
// step budget ran out, [4..5] one uint64: decoder evaluations of the march (speculative samples included), [6] hits, [7] rays handed to the
// looping kernel's second stage.
// While the device-side count is >= tail_rows a step is two launches: the decoder on the active rows (64- / 128-row tiles, MFMA-bound)
// and sdfr_trace_step_kernel (advance, retire, ballot compaction).  Once it drops below tail_rows -- every 16-ray tile then has a CU to
// itself -- ONE launch of the decoder kernel in MODE 4 takes the remaining rays to termination: the workgroup loops over decoder pass ->
// step rule -> hit / exit test for its 16 rays with the ray state in registers (no per-step launch, no compaction, no host read).  The gate
// is evaluated on the device in every step of the head; after `head_steps` steps an unconditional tail launch takes whatever is left.
// Speculative passes: `levels` (host array int32[n_levels][2] = first pass index, samples per ray and pass; both ascending; samples 4, 8, 16, 32
// or 64) is the march's schedule -- from pass index levels[i][0] on a pass evaluates levels[i][1] samples per ray (p_0 = lam, p_j = p_{j-1} +
// sigma q^j rho / |d|, q = ratio of the ray's last two radii clamped to [0.5, q_max]) and accepts the prefix that stays inside the previous
// samples' safe spheres.  A 64-row pass of the looping kernel costs what a 16-row pass costs (both are paced by the weight stream through the
// CU), so the rows a tile carries are spent on ever fewer rays: 64 / k rays per tile at level k.  What keeps a march alive are the rays that
// creep along the surface at grazing incidence and MISS (256x256 bench crop: every hit is found by pass 10, the last miss leaves the cube in
// pass 15), and they advance k samples per pass.  The pass index alone decides (head_steps is clamped to the first level's pass), so a ray's
// sample sequence does not depend on the launch schedule or on the batch.
// Every level is one launch of the looping kernel (a STAGE): a pool of persistent workgroups fetches tiles from the stage's device counter
// (counters[16 + stage]), marches each through the stage's passes and appends the survivors to the next stage's list (counters[7 + stage];
// lists: pix2 / lam2 and pix0 / lam0 in turn -- the head's lists are free by then).  r04: up to SDFR_TRACE_LEVELS levels (r03: two, fixed
// tiles-per-ray grids and a scratch tile per ray tile).
// tail_rows_buf: scratch float[SDFR_TRACE_POOL][64][L + 3] (one tile of operand rows per pool workgroup).
#define SDFR_TRACE_POOL 1024
extern "C" int sdfr_trace_pool(void) { return SDFR_TRACE_POOL; }
static int trace_march_impl(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D,
                            float eps, int steps, int head_steps, int tail_rows, const int32_t* levels, int n_levels, float q_max,
                            float sigma, int half, int32_t* counters, int32_t* pix0, float* lam0_, int32_t* pix1, float* lam1_,
                            int32_t* pix2, float* lam2_, const float* far, float* inputs, float* sdf, float* tail_rows_buf, float* hit_lam,
                            float* hit_sdf, void* stream);
extern "C" int sdfr_trace_march(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H,
                                float eps, int steps, int head_steps, int tail_rows, const int32_t* levels, int n_levels, float q_max,
                                float sigma, int half, int32_t* counters, int32_t* pix0, float* lam0_, int32_t* pix1, float* lam1_,
                                int32_t* pix2, float* lam2_, const float* far, float* inputs, float* sdf, float* tail_rows_buf, float* hit_lam,
                                float* hit_sdf, void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0, "sdfr_trace_march: bad size");
    return trace_march_impl(d, pose, Kinv, latn, L, B, dense_dims(W, H, 0), eps, steps, head_steps, tail_rows, levels, n_levels, q_max, sigma,
                            half, counters, pix0, lam0_, pix1, lam1_, pix2, lam2_, far, inputs, sdf, tail_rows_buf, hit_lam, hit_sdf, stream);
}
extern "C" int sdfr_trace_march_r(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext,
                                  float eps, int steps, int head_steps, int tail_rows, const int32_t* levels, int n_levels, float q_max,
                                  float sigma, int half, int32_t* counters, int32_t* pix0, float* lam0_, int32_t* pix1, float* lam1_,
                                  int32_t* pix2, float* lam2_, const float* far, float* inputs, float* sdf, float* tail_rows_buf, float* hit_lam,
                                  float* hit_sdf, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_march_r");
    if (rc) return rc;
    return trace_march_impl(d, pose, Kinv, latn, L, B, D, eps, steps, head_steps, tail_rows, levels, n_levels, q_max, sigma, half, counters,
                            pix0, lam0_, pix1, lam1_, pix2, lam2_, far, inputs, sdf, tail_rows_buf, hit_lam, hit_sdf, stream);
}
static int trace_march_impl(const sdfr_decoder* d, const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D,
                            float eps, int steps, int head_steps, int tail_rows, const int32_t* levels, int n_levels, float q_max,
                            float sigma, int half, int32_t* counters, int32_t* pix0, float* lam0_, int32_t* pix1, float* lam1_,
                            int32_t* pix2, float* lam2_, const float* far, float* inputs, float* sdf, float* tail_rows_buf, float* hit_lam,
                            float* hit_sdf, void* stream) {
    SDFR_REQUIRE(d && pose && Kinv && latn && counters && pix0 && lam0_ && pix1 && lam1_ && far && inputs && sdf && tail_rows_buf && hit_lam &&
                     hit_sdf, "sdfr_trace_march: NULL argument");
    float4* lam0 = reinterpret_cast<float4*>(lam0_);
    float4* lam1 = reinterpret_cast<float4*>(lam1_);
    float4* lam2 = reinterpret_cast<float4*>(lam2_);
    SDFR_REQUIRE(B > 0 && steps > 0 && head_steps >= 0 && tail_rows >= 0, "sdfr_trace_march: bad size");
    SDFR_REQUIRE(n_levels >= 0 && n_levels <= SDFR_TRACE_LEVELS && (n_levels == 0 || levels), "sdfr_trace_march: 0 ... %d speculation levels",
                 SDFR_TRACE_LEVELS);
    SDFR_REQUIRE(q_max >= 1.f && q_max <= 4.f, "sdfr_trace_march: q_max = %g (1 ... 4)", (double)q_max);
    for (int i = 0; i < n_levels; ++i) {
        const int k = levels[2 * i + 1];
        SDFR_REQUIRE(k == 4 || k == 8 || k == 16 || k == 32 || k == 64, "sdfr_trace_march: level %d takes %d samples per pass (4, 8, 16, 32 or 64)", i, k);
        SDFR_REQUIRE(levels[2 * i] >= 0 && (i == 0 || (levels[2 * i] > levels[2 * i - 2] && k > levels[2 * i - 1])),
                     "sdfr_trace_march: levels must ascend in pass index and in samples");
    }
    SDFR_REQUIRE(d->n_inputs == L + 3, "sdfr_trace_march: decoder with L + 3 = %d inputs expected, it has %d", L + 3, d->n_inputs);
    const int64_t n_max = (int64_t)B * D.PS;
    SDFR_REQUIRE(n_max < (int64_t)1 << 31, "sdfr_trace_march: too many rays");
    hipStream_t s = (hipStream_t)stream;
    if (d->HP != 512 || d->has_ln) {
        // LayerNorm decoders and hidden widths below 257: the looping kernel (MODE 4) is built for the 512-wide weight-norm / plain decoders
        // only; these march with per-step launches of their own forward kernels (device-side count, float32, plain sphere tracing -- the
        // oracle's arithmetic without speculation), all `steps` of them: correct, not tuned
        SDFR_REQUIRE(!half, "sdfr_trace_march: half operands need a 512-wide decoder without LayerNorm");
        unsigned long long* evals_ = reinterpret_cast<unsigned long long*>(counters + 4);
        for (int step = 0; step < steps; ++step) {
            int rc = sdfr_mlp_forward_counted(d, inputs, n_max, counters + step % 3, sdf, 0, stream);
            if (rc != SDFR_OK) return rc;
            const int a = step & 1;
            hipLaunchKernelGGL(sdfr_trace_step_kernel, dim3(sdfr_cdiv(n_max, 256)), dim3(256), 0, s, pose, Kinv, latn, L, D, eps, sdf,
                               counters + step % 3, counters + (step + 1) % 3, counters + (step + 2) % 3, a ? pix1 : pix0,
                               a ? reinterpret_cast<float4*>(lam1_) : reinterpret_cast<float4*>(lam0_), a ? pix0 : pix1,
                               a ? reinterpret_cast<float4*>(lam0_) : reinterpret_cast<float4*>(lam1_), far, inputs, hit_lam, hit_sdf, 1, evals_, q_max);
        }
        hipLaunchKernelGGL(sdfr_trace_leftover_kernel, dim3(1), dim3(64), 0, s, counters + steps % 3, counters + 3);
        SDFR_LAUNCH_CHECK();
        return SDFR_OK;
    }
    // levels that start inside the step budget
    int nlv = 0;
    while (nlv < n_levels && levels[2 * nlv] < steps) ++nlv;
    const int spec_from = nlv > 0 ? levels[0] : 0x7fffffff;
    const int spec_k = nlv > 0 ? 4 : 1;                          // which looping kernel: 64-row tiles (speculative schedules) or 16-row tiles (plain)
    SDFR_REQUIRE(nlv <= 1 || (pix2 && lam2), "sdfr_trace_march: more than one speculation level needs the hand-over list (pix2, lam2)");
    if (head_steps > steps) head_steps = steps;
    if (head_steps > spec_from) head_steps = spec_from;          // speculative passes exist in the looping kernel only
    unsigned long long* evals = reinterpret_cast<unsigned long long*>(counters + 4);
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n_max; P.sdf = sdf; P.maskbuf = nullptr; P.trace = nullptr;
    P.t_far = far; P.t_pose = pose; P.t_Kinv = Kinv; P.t_latn = latn; P.t_hit_lam = hit_lam; P.t_hit_sdf = hit_sdf; P.t_W = D.W; P.t_H = D.H; P.t_wh = D.wh; P.t_PS = D.PS;
    P.t_eps = eps; P.t_sigma = sigma; P.t_qmax = q_max; P.t_evals = evals; P.t_unresolved = counters + 3;
    P.t_nlv = nlv;
    for (int i = 0; i < SDFR_TRACE_LEVELS; ++i) { P.t_lv_from[i] = i < nlv ? levels[2 * i] : 0x7fffffff; P.t_lv_k[i] = i < nlv ? levels[2 * i + 1] : 1; }
    auto launch_tail = [&](const MlpParams& T) {
        // (a launch gated to counts below n_dev_hi needs workgroups for that many rays only; beyond SDFR_TRACE_POOL tiles the pool's workgroups
        // fetch one tile after the other)
        const int64_t n_grid = n_max < (int64_t)T.n_dev_hi ? n_max : (int64_t)T.n_dev_hi;
        const int64_t pool = (int64_t)SDFR_TRACE_POOL * T.t_rt;
        if (half) sdfr_launch_tail_f16_512(T, n_grid < pool ? n_grid : pool, spec_k, s); else sdfr_launch_tail_f32_512(T, n_grid < pool ? n_grid : pool, spec_k, s);
    };
    // stage 0: from pass `step` (whichever step the device-side count picks, or the end of the head) to the second level's first pass or the end
    // of the budget: 64 / levels[0][1] rays per tile (plain schedules: 16 rays on 16 rows)
    auto stage0 = [&](int step, int hi) {
        MlpParams T = P;
        T.inputs = tail_rows_buf; T.t_rows = tail_rows_buf; T.t_rt = nlv > 0 ? 64 / levels[1] : 16;
        T.n_dev = counters + step % 3; T.n_dev_lo = 1; T.n_dev_hi = hi;
        T.t_pix = (step & 1) ? pix1 : pix0; T.t_lam = (step & 1) ? lam1 : lam0; T.t_steps = steps - step; T.t_step0 = step;
        T.t_stage = T.t_steps; T.t_next_cnt = nullptr; T.t_next_pix = nullptr; T.t_next_lam = nullptr;
        T.t_tile_ctr = counters + 16;
        if (nlv > 1 && step < levels[2]) { T.t_stage = levels[2] - step; T.t_next_cnt = counters + 7; T.t_next_pix = pix2; T.t_next_lam = lam2; }
        launch_tail(T);
    };
    for (int step = 0; step < head_steps; ++step) {
        MlpParams F = P;
        F.n_dev = counters + step % 3; F.n_dev_lo = tail_rows > 0 ? tail_rows : 1; F.n_dev_hi = 0x7fffffff;
        if (half) {
            // two tile sizes, picked by the device-side count: 128-row tiles while they fill the chip at least once (>= 64 rows x 256 CUs
            // would already do with the smaller tile), 64-row tiles below -- one pass of a half-size tile per step instead of a full-size one
            const int mid = 64 * 256;
            if (F.n_dev_lo < mid) {
                MlpParams M = F;
                M.n_dev_hi = mid;
                sdfr_launch_fwd_f16_512_tile64(M, n_max < mid ? n_max : (int64_t)mid, s);
                F.n_dev_lo = mid;
            }
            if (n_max >= F.n_dev_lo) sdfr_launch_fwd_f16_512(F, n_max, false, s);
        } else sdfr_launch_fwd_f32_512(F, n_max, false, s);
        if (tail_rows > 0) stage0(step, tail_rows);
        const int a = step & 1;
        hipLaunchKernelGGL(sdfr_trace_step_kernel, dim3(sdfr_cdiv(n_max, 256)), dim3(256), 0, s, pose, Kinv, latn, L, D, eps, sdf,
                           counters + step % 3, counters + (step + 1) % 3, counters + (step + 2) % 3, a ? pix1 : pix0, a ? lam1 : lam0,
                           a ? pix0 : pix1, a ? lam0 : lam1, far, inputs, hit_lam, hit_sdf, tail_rows > 0 ? tail_rows : 1, evals, q_max);
    }
    if (head_steps < steps) {
        stage0(head_steps, 0x7fffffff);
        // stages 1 ...: the survivors of the stage before (count in counters[6 + i]), 64 / k rays per tile, to the next level's first pass or the end
        for (int i = 1; i < nlv; ++i) {
            MlpParams T = P;
            T.inputs = tail_rows_buf; T.t_rows = tail_rows_buf; T.t_rt = 64 / levels[2 * i + 1];
            T.n_dev = counters + 6 + i; T.n_dev_lo = 1; T.n_dev_hi = 0x7fffffff;
            T.t_pix = (i & 1) ? pix2 : pix0; T.t_lam = (i & 1) ? lam2 : lam0; T.t_steps = steps - levels[2 * i]; T.t_step0 = levels[2 * i];
            T.t_stage = T.t_steps; T.t_next_cnt = nullptr; T.t_next_pix = nullptr; T.t_next_lam = nullptr;
            T.t_tile_ctr = counters + 16 + i;
            if (i + 1 < nlv) {
                T.t_stage = levels[2 * i + 2] - levels[2 * i]; T.t_next_cnt = counters + 7 + i;
                T.t_next_pix = (i & 1) ? pix0 : pix2; T.t_next_lam = (i & 1) ? lam0 : lam2;
            }
            launch_tail(T);
        }
    } else {
        // the step budget ended in the head: the rays still listed are unresolved
        hipLaunchKernelGGL(sdfr_trace_leftover_kernel, dim3(1), dim3(64), 0, s, counters + steps % 3, counters + 3);
    }
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}


// hit list + rows for the Jacobian pass.  n_hits = counters + 6 (zeroed by sdfr_trace_setup).
static int trace_hits_impl(const float* pose, const float* Kinv, const float* latn, int L, int B, const TraceDims& D, const float* hit_lam,
                           int32_t* n_hits, int32_t* hit_slot, int32_t* idx, float* rows, void* stream) {
    SDFR_REQUIRE(pose && Kinv && latn && hit_lam && n_hits && hit_slot && idx && rows, "sdfr_trace_hits: NULL argument");
    SDFR_REQUIRE(L >= 0 && B > 0, "sdfr_trace_hits: bad size");
    hipLaunchKernelGGL(sdfr_trace_hits_kernel, dim3(sdfr_cdiv((int64_t)D.PS, 256), B), dim3(256), 0, (hipStream_t)stream, pose, Kinv, latn, L, D,
                       hit_lam, n_hits, hit_slot, idx, rows);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
extern "C" int sdfr_trace_hits(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, const float* hit_lam,
                               int32_t* n_hits, int32_t* hit_slot, int32_t* idx, float* rows, void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0, "sdfr_trace_hits: bad size");
    return trace_hits_impl(pose, Kinv, latn, L, B, dense_dims(W, H, 0), hit_lam, n_hits, hit_slot, idx, rows, stream);
}
extern "C" int sdfr_trace_hits_r(const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext, const float* hit_lam,
                                 int32_t* n_hits, int32_t* hit_slot, int32_t* idx, float* rows, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_hits_r");
    if (rc) return rc;
    return trace_hits_impl(pose, Kinv, latn, L, B, D, hit_lam, n_hits, hit_slot, idx, rows, stream);
}

static int trace_composite_impl(const float* pose, const float* Kinv, int L, int B, const TraceDims& D, const float* hit_lam, const int32_t* hit_slot,
                                const float* J, const float* f0, float* color, float* mask, float* depth, float* normals, float* lam_s,
                                void* stream) {
    SDFR_REQUIRE(pose && Kinv && hit_lam && hit_slot && J && f0 && color && mask && depth && normals, "sdfr_trace_composite: NULL argument");
    SDFR_REQUIRE(L >= 0 && B > 0, "sdfr_trace_composite: bad size");
    hipLaunchKernelGGL(sdfr_trace_composite_kernel, dim3(sdfr_cdiv((int64_t)D.PS, 256), B), dim3(256), 0, (hipStream_t)stream, pose, Kinv, L, D,
                       hit_lam, hit_slot, J, f0, color, mask, depth, normals, lam_s);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
extern "C" int sdfr_trace_composite(const float* pose, const float* Kinv, int L, int B, int W, int H, const float* hit_lam, const int32_t* hit_slot,
                                    const float* J, const float* f0, float* color, float* mask, float* depth, float* normals, float* lam_s,
                                    void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0, "sdfr_trace_composite: bad size");
    return trace_composite_impl(pose, Kinv, L, B, dense_dims(W, H, 0), hit_lam, hit_slot, J, f0, color, mask, depth, normals, lam_s, stream);
}
extern "C" int sdfr_trace_composite_r(const float* pose, const float* Kinv, int L, int B, const sdfr_extents* ext, const float* hit_lam,
                                      const int32_t* hit_slot, const float* J, const float* f0, float* color, float* mask, float* depth,
                                      float* normals, float* lam_s, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_composite_r");
    if (rc) return rc;
    return trace_composite_impl(pose, Kinv, L, B, D, hit_lam, hit_slot, J, f0, color, mask, depth, normals, lam_s, stream);
}

extern "C" int64_t sdfr_trace_backward_ws_floats(int B, int W, int H) {
    return (int64_t)B * sdfr_cdiv((int64_t)W * H, TRB_THREADS) * (12 + TRB_MAXL);
}

// image gradients (any of them may be NULL) -> g_pose [B][16] (row-major 4x4: rotation and translation entries) and g_latn [B][L] (gradient
// w.r.t. the NORMALISED latent); sdfr_params_backward turns them into the gradients of yaw, trans and the latent.
static int trace_refine_backward_impl(const float* pose, const float* Kinv, int L, int B, const TraceDims& D, const float* hit_lam,
                                      const int32_t* hit_slot, const float* J, const float* f0, const float* g_color, const float* g_depth,
                                      const float* g_normals, const float* g_xyzf, const int32_t* pt_slot, int ecap, int surfel, float* ws,
                                      float* g_pose, float* g_latn, void* stream) {
    SDFR_REQUIRE(pose && Kinv && hit_lam && hit_slot && J && f0 && ws && g_pose && g_latn, "sdfr_trace_refine_backward: NULL argument");
    SDFR_REQUIRE(L >= 0 && L <= TRB_MAXL && B > 0, "sdfr_trace_refine_backward: latent size 0..%d", TRB_MAXL);
    SDFR_REQUIRE(!g_xyzf || (pt_slot && ecap > 0), "sdfr_trace_refine_backward: g_xyzf needs pt_slot and ecap (sdfr_trace_points)");
    const int nblk = sdfr_cdiv((int64_t)D.PS, TRB_THREADS);
    hipLaunchKernelGGL(sdfr_trace_backward_kernel, dim3(nblk, B), dim3(TRB_THREADS), 0, (hipStream_t)stream, pose, Kinv, L, D, hit_lam, hit_slot,
                       J, f0, g_color, g_depth, g_normals, g_xyzf, pt_slot, ecap, surfel, ws);
    hipLaunchKernelGGL(sdfr_trace_backward_sum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ws, nblk, L, g_pose, g_latn);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
extern "C" int sdfr_trace_refine_backward(const float* pose, const float* Kinv, int L, int B, int W, int H, const float* hit_lam,
                                          const int32_t* hit_slot, const float* J, const float* f0, const float* g_color, const float* g_depth,
                                          const float* g_normals, const float* g_xyzf, const int32_t* pt_slot, int ecap, int surfel, float* ws,
                                          float* g_pose, float* g_latn, void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0, "sdfr_trace_refine_backward: bad size");
    return trace_refine_backward_impl(pose, Kinv, L, B, dense_dims(W, H, 0), hit_lam, hit_slot, J, f0, g_color, g_depth, g_normals, g_xyzf, pt_slot, ecap,
                                      surfel, ws, g_pose, g_latn, stream);
}
extern "C" int sdfr_trace_refine_backward_r(const float* pose, const float* Kinv, int L, int B, const sdfr_extents* ext, const float* hit_lam,
                                            const int32_t* hit_slot, const float* J, const float* f0, const float* g_color, const float* g_depth,
                                            const float* g_normals, const float* g_xyzf, const int32_t* pt_slot, int ecap, int surfel, float* ws,
                                            float* g_pose, float* g_latn, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_refine_backward_r");
    if (rc) return rc;
    return trace_refine_backward_impl(pose, Kinv, L, B, D, hit_lam, hit_slot, J, f0, g_color, g_depth, g_normals, g_xyzf, pt_slot, ecap, surfel, ws,
                                      g_pose, g_latn, stream);
}

extern "C" int sdfr_trace_backward(const float* pose, const float* Kinv, int L, int B, int W, int H, const float* hit_lam, const int32_t* hit_slot,
                                   const float* J, const float* f0, const float* g_color, const float* g_depth, const float* g_normals,
                                   float* ws, float* g_pose, float* g_latn, void* stream) {
    return sdfr_trace_refine_backward(pose, Kinv, L, B, W, H, hit_lam, hit_slot, J, f0, g_color, g_depth, g_normals, nullptr, nullptr, 0, 0, ws,
                                      g_pose, g_latn, stream);
}

// points['xyzf'] of a traced render (what the refinement loop's 3-D loss consumes, pipelines/optimizer.py:125-130): the camera-frame hit
// points p_cam = lam_s K^-1 [x, y, 1] of a crop's hit pixels, compacted in PIXEL ORDER (row-major; deterministic, unlike the hit list of
// sdfr_trace_hits) into xyzf [B][ecap][3]; ecnt [B] = the crop's TRUE hit count (callers compare against ecap; surplus rows are dropped);
// pt_slot [B*W*H] = a pixel's row or -1.  One workgroup per crop: ballot + running offset.
#define TRP_THREADS 1024
__global__ __launch_bounds__(TRP_THREADS) void sdfr_trace_points_kernel(const float* __restrict__ Kinv, const TraceDims D,
                                                                       const int32_t* __restrict__ hit_slot, const float* __restrict__ lam_s,
                                                                       float* __restrict__ xyzf, int ecap, int32_t* __restrict__ ecnt,
                                                                       int32_t* __restrict__ pt_slot) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int W, H;
    trace_dims(D, b, W, H);
    const int P_ = W * H, PS = D.PS;
    const float* Ki = Kinv + (int64_t)b * 9;
    __shared__ int wcnt[TRP_THREADS / 64];
    __shared__ int base;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int p0 = 0; p0 < P_; p0 += TRP_THREADS) {
        const int p = p0 + tid;
        const int64_t gp = (int64_t)b * PS + p;
        const bool hit = p < P_ && hit_slot[gp] >= 0;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (p < P_) pt_slot[gp] = (hit && pos < ecap) ? pos : -1;
        if (hit && pos < ecap) {
            const float x = (float)(p % W), y = (float)(p / W), l = lam_s[gp];
            float* o = xyzf + ((int64_t)b * ecap + pos) * 3;
            o[0] = l * (fmaf(Ki[1], y, Ki[0] * x) + Ki[2]);
            o[1] = l * (fmaf(Ki[4], y, Ki[3] * x) + Ki[5]);
            o[2] = l * (fmaf(Ki[7], y, Ki[6] * x) + Ki[8]);
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < TRP_THREADS / 64; ++w) t += wcnt[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0) ecnt[b] = base;
}

static int trace_points_impl(const float* Kinv, int B, const TraceDims& D, const int32_t* hit_slot, const float* lam_s, float* xyzf, int ecap,
                             int32_t* ecnt, int32_t* pt_slot, void* stream);
extern "C" int sdfr_trace_points(const float* Kinv, int B, int W, int H, const int32_t* hit_slot, const float* lam_s, float* xyzf, int ecap,
                                 int32_t* ecnt, int32_t* pt_slot, void* stream) {
    SDFR_REQUIRE(W > 0 && H > 0, "sdfr_trace_points: bad size");
    return trace_points_impl(Kinv, B, dense_dims(W, H, 0), hit_slot, lam_s, xyzf, ecap, ecnt, pt_slot, stream);
}
extern "C" int sdfr_trace_points_r(const float* Kinv, int B, const sdfr_extents* ext, const int32_t* hit_slot, const float* lam_s, float* xyzf, int ecap,
                                   int32_t* ecnt, int32_t* pt_slot, void* stream) {
    TraceDims D;
    int rc = ragged_dims(D, ext, "sdfr_trace_points_r");
    if (rc) return rc;
    return trace_points_impl(Kinv, B, D, hit_slot, lam_s, xyzf, ecap, ecnt, pt_slot, stream);
}
static int trace_points_impl(const float* Kinv, int B, const TraceDims& D, const int32_t* hit_slot, const float* lam_s, float* xyzf, int ecap,
                             int32_t* ecnt, int32_t* pt_slot, void* stream) {
    SDFR_REQUIRE(Kinv && hit_slot && lam_s && xyzf && ecnt && pt_slot, "sdfr_trace_points: NULL argument");
    SDFR_REQUIRE(B > 0 && ecap > 0, "sdfr_trace_points: bad size");
    hipLaunchKernelGGL(sdfr_trace_points_kernel, dim3(B), dim3(TRP_THREADS), 0, (hipStream_t)stream, Kinv, D, hit_slot, lam_s, xyzf, ecap, ecnt,
                       pt_slot);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
