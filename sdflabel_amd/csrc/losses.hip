// Losses and solver step of the reference's refinement loop, on the device and for B crops at once.
//
//   sdfr_loss_3d      compute_loss_3d (pipelines/optimizer.py:166-198): nearest lidar point of every front-facing estimated point
//                     (the reference round-trips to the host and builds a sklearn KDTree per iteration, :180-181; here an exact
//                     brute-force search over LDS tiles), pairs closer than threshold/scale, mean pair distance, and the
//                     gradients w.r.t. the estimated points and the scale (the lidar cloud is divided by scale, :84).
//   sdfr_loss_2d      compute_loss_2d (:200-237): for every rendered (non-zero) pixel the smallest NOCS distance to the target
//                     weighted by clamp(diam - pixel distance, 0).  The reference forms Q x H x W tensors (~100 GB at 256^2);
//                     algebraically this is a min over the (2*diam-1)^2 window plus the constant ||rendered|| of every pixel
//                     outside the window, evaluated here per pixel.
//   sdfr_solver_step  the MultipleOptimizer step (:13-23,44-52): Adam(lr .01) on yaw and trans, SGD(lr .01 / 3e-5) on scale / latent,
//                     gated per crop by the loop's skip conditions (:127-129,149-151).
// Fixed-order reductions throughout: bit-repeatable.  Compiled with -ffp-contract=off.
#include "sdfr_common.h"
#include "solver.h"
#include <float.h>

// ---- 3-D loss ----------------------------------------------------------------------------------------------------------
// pass 1: one workgroup of 4 waves per 64 estimated points (lane = point); wave w scans lidar points [w*nl/4, (w+1)*nl/4) out of an LDS
//         tile, the four candidates are merged in wave order with a strict '<' (so ties resolve to the lowest lidar index, as a
//         sequential scan would); the point's un-normalised gradient and the workgroup's partial sums (distance, d/dscale, pairs) are
//         written in a fixed order;
// pass 2: every workgroup re-reduces the partials in the same order and scales its share of the gradient; workgroup 0 writes the loss.
#define L3_PTS 64
#define L3_NW 4
#define L3_TILE 1024
// (the body of the pairs pass as a device function of (block, crop, blocks per crop): shared by the kernel below and by the fused launch of
// both losses, sdfr_losses_fused_kernel)
__device__ __forceinline__ void loss_3d_pairs_block(const int blk, const int b, const int nblk, const float* __restrict__ est,
                                                    const int32_t* __restrict__ ecnt, int ecap, const float* __restrict__ lidar,
                                                    const int32_t* __restrict__ lcnt, int lcap, const float* __restrict__ scale, float threshold,
                                                    float* __restrict__ g_est, float* __restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ne = sdfr_count(ecnt, b, ecap), nl = sdfr_count(lcnt, b, lcap);
    const float s = scale[b];
    const float thr = threshold / s;                                   // :184
    __shared__ float tile[3][L3_TILE];
    __shared__ float cd[L3_NW][L3_PTS];
    __shared__ int ci[L3_NW][L3_PTS];
    const float* E = est + (int64_t)b * ecap * 3;
    const float* Lp = lidar + (int64_t)b * lcap * 3;
    float* G = g_est + (int64_t)b * ecap * 3;
    const int j = blk * L3_PTS + lane;
    if (blk * L3_PTS >= ne || nl == 0) {                        // nothing to pair in this workgroup: zero rows, zero partials
        if (wave == 0) {
            if (j < ecap) { G[j * 3] = 0.f; G[j * 3 + 1] = 0.f; G[j * 3 + 2] = 0.f; }
            if (lane < 3) partial[((int64_t)b * nblk + blk) * 3 + lane] = 0.f;
        }
        return;
    }
    const bool act = j < ne;
    const float ex = act ? E[j * 3] : 0.f, ey = act ? E[j * 3 + 1] : 0.f, ez = act ? E[j * 3 + 2] : 0.f;
    float best = FLT_MAX;
    int bi = -1;
    for (int m0 = 0; m0 < nl; m0 += L3_TILE) {
        __syncthreads();
        for (int m = tid; m < L3_TILE && m0 + m < nl; m += 64 * L3_NW) {      // lidar / scale (:84), staged once per tile
            tile[0][m] = Lp[(m0 + m) * 3] / s; tile[1][m] = Lp[(m0 + m) * 3 + 1] / s; tile[2][m] = Lp[(m0 + m) * 3 + 2] / s;
        }
        __syncthreads();
        const int mn = min(L3_TILE, nl - m0);
        const int q = (mn + L3_NW - 1) / L3_NW;
        const int lo = wave * q, hi = min(mn, lo + q);
        for (int m = lo; m < hi; ++m) {
            const float dx = tile[0][m] - ex, dy = tile[1][m] - ey, dz = tile[2][m] - ez;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bi = m0 + m; }
        }
    }
    cd[wave][lane] = best; ci[wave][lane] = bi;
    __syncthreads();
    float lsum = 0.f, gs = 0.f, cnt = 0.f;
    if (wave == 0) {
        // the waves scanned ascending, disjoint index ranges per tile; over several tiles a later tile may hold a smaller index range of
        // another wave, so ties are broken on the index explicitly
#pragma unroll
        for (int w = 1; w < L3_NW; ++w) {
            const float d2 = cd[w][lane];
            const int i2 = ci[w][lane];
            if (d2 < best || (d2 == best && i2 >= 0 && i2 < bi)) { best = d2; bi = i2; }
        }
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (act && bi >= 0 && sqrtf(best) < thr) {                       // :184
            const float lx = Lp[bi * 3] / s, ly = Lp[bi * 3 + 1] / s, lz = Lp[bi * 3 + 2] / s;
            const float dx = lx - ex, dy = ly - ey, dz = lz - ez;
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);          // :185
            lsum = d;
            cnt = 1.f;
            if (d > 0.f) {
                const float ux = dx / d, uy = dy / d, uz = dz / d;       // d||l/s - e|| / d(l/s)
                gx = -ux; gy = -uy; gz = -uz;
                gs = -(ux * lx + uy * ly + uz * lz) / s;                 // d(l/s)/ds = -(l/s)/s
            }
        }
        if (j < ecap) { G[j * 3] = gx; G[j * 3 + 1] = gy; G[j * 3 + 2] = gz; }     // rows beyond ne: zero
        float v[3] = {lsum, gs, cnt};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float x = v[i];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if (lane == 0) partial[((int64_t)b * nblk + blk) * 3 + i] = x;
        }
    }
}

__global__ __launch_bounds__(64 * L3_NW) void sdfr_loss_3d_pairs_kernel(const float* __restrict__ est, const int32_t* __restrict__ ecnt,
                                                                       int ecap, const float* __restrict__ lidar,
                                                                       const int32_t* __restrict__ lcnt, int lcap,
                                                                       const float* __restrict__ scale, float threshold,
                                                                       float* __restrict__ g_est, float* __restrict__ partial) {
    loss_3d_pairs_block(blockIdx.x, blockIdx.y, gridDim.x, est, ecnt, ecap, lidar, lcnt, lcap, scale, threshold, g_est, partial);
}

__global__ __launch_bounds__(256) void sdfr_loss_3d_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                   const int32_t* __restrict__ ecnt, int ecap,
                                                                   const int32_t* __restrict__ lcnt, int lcap, float weight,
                                                                   float* __restrict__ loss, float* __restrict__ g_est,
                                                                   float* __restrict__ g_scale, int32_t* __restrict__ npairs) {
    const int b = blockIdx.y, tid = threadIdx.x;
    __shared__ float red[3][256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = tid; i < nblk; i += 256) {
        const float* p = partial + ((int64_t)b * nblk + i) * 3;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; red[2][tid] += red[2][tid + st]; }
        __syncthreads();
    }
    const float tot = red[0][0], gst = red[1][0], cf = red[2][0];
    const float inv = cf > 0.f ? 1.f / cf : 0.f;
    const int64_t e = (int64_t)blockIdx.x * 256 + tid;                   // mean over the pairs (:189), times the loss weight
    if (e < (int64_t)3 * ecap) g_est[(int64_t)b * 3 * ecap + e] *= weight * inv;
    if (blockIdx.x == 0 && tid == 0) {
        const int ne = sdfr_count(ecnt, b, ecap), nl = sdfr_count(lcnt, b, lcap);
        loss[b] = cf > 0.f ? tot * inv : 0.f;                            // :188-191
        g_scale[b] = weight * gst * inv;
        npairs[b] = (ne > 0 && nl > 0) ? (int)cf : -1;                   // -1: a cloud is empty -> the loop skips the crop (:127-129)
    }
}

extern "C" int sdfr_loss_3d(const float* est, const int32_t* ecnt, int ecap, const float* lidar, const int32_t* lcnt, int lcap,
                            const float* scale, float threshold, float weight, int B, float* loss, float* g_est, float* g_scale,
                            int32_t* npairs, float* scratch, void* stream) {
    SDFR_REQUIRE(est && lidar && scale && loss && g_est && g_scale && npairs && scratch, "sdfr_loss_3d: NULL argument");
    SDFR_REQUIRE(ecap > 0 && lcap >= 0, "sdfr_loss_3d: bad capacity");
    if (B <= 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = sdfr_cdiv(ecap, L3_PTS);
    hipLaunchKernelGGL(sdfr_loss_3d_pairs_kernel, dim3(nblk, B), dim3(64 * L3_NW), 0, s, est, ecnt, ecap, lidar, lcnt, lcap, scale, threshold,
                       g_est, scratch);
    SDFR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sdfr_loss_3d_finalize_kernel, dim3(sdfr_cdiv(3 * ecap, 256), B), dim3(256), 0, s, scratch, nblk, ecnt, ecap, lcnt, lcap,
                       weight, loss, g_est, g_scale, npairs);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- 2-D loss ----------------------------------------------------------------------------------------------------------
// pass 1: one thread per pixel, one workgroup per 16x16 pixel tile.  A tile that holds a rendered pixel stages the target window
//         (tile + halo, all three channels) and the (2*rad+1)^2 tap weights in LDS; the window search then runs out of LDS in the
//         reference's tap order.  Un-normalised gradient per pixel, fixed-order partial sums per tile;
// pass 2: every workgroup re-reduces the partials in the same order, scales its share of the gradient; workgroup 0 writes the loss.
#define L2_T 16
#define L2_RMAX 8                       // window radius the LDS path is built for (diam <= 9); larger windows read the target from memory
#define L2_S 48                         // LDS row pitch: consecutive tile rows fall 16 banks apart
template <bool LDS, int RADC>       // RADC > 0: window radius known at compile time (the loops unroll and the LDS reads of a row batch up); LDS: window radius <= L2_RMAX, taps come from the staged tile (two instantiations: a run-time choice per tap would
                          // turn the loads into flat accesses with a full wait after each)
__device__ __forceinline__ void loss_2d_pixels_block(const int blk, const int b, const int nblk, const float* __restrict__ rend,
                                                     const float* __restrict__ target, int H, int W, const int32_t* __restrict__ wh, int pst,
                                                     float diam, float threshold_nocs, float* __restrict__ g_rend, float* __restrict__ partial) {
    const int tid = threadIdx.x;
    int P = H * W;                                                       // pixel stride of the image channels
    if (wh) {
        // ragged extents (r04): crop b is W_b x H_b pixels in a slot of pst pixels per channel; the launch covers the largest tile count, the
        // tiles beyond this crop's contribute exact zeros (the fixed-order sums below then equal the crop's own launch bit for bit)
        W = wh[2 * b]; H = wh[2 * b + 1]; P = pst;
        if (W < 1 || H < 1 || (int64_t)W * H > (int64_t)pst) { W = 0; H = 0; }          // outside the contract: an empty crop (zero loss)
        if ((int)blk >= ((W + L2_T - 1) / L2_T) * ((H + L2_T - 1) / L2_T)) {
            if (tid < 3) partial[((int64_t)b * nblk + blk) * 3 + tid] = 0.f;
            return;
        }
    }
    const float* R = rend + (int64_t)b * 3 * P;
    const float* Tg = target + (int64_t)b * 3 * P;
    float* G = g_rend + (int64_t)b * 3 * P;
    constexpr int UNR = RADC > 0 ? 2 * RADC + 1 : 1;                    // full unroll of the window loops when the radius is a constant
    const int rad = RADC > 0 ? RADC : (int)ceilf(diam) - 1;             // taps with clamp(diam - dist, 0) > 0 have |d| < diam
    const int tilesX = (W + L2_T - 1) / L2_T;
    const int tx = blk % tilesX, ty = blk / tilesX;
    const int lx = tid & (L2_T - 1), ly = tid / L2_T;
    const int w = tx * L2_T + lx, h = ty * L2_T + ly;
    const bool inside = (w < W) && (h < H);
    const int q = h * W + w;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (inside) { r0 = R[q]; r1 = R[P + q]; r2 = R[2 * P + q]; }
    const bool nz = inside && (r0 + r1 + r2 != 0.f);                    // rendering_nocs.sum(0).nonzero()  (:213)
    __shared__ float tg[3][(L2_T + 2 * L2_RMAX) * L2_S];
    __shared__ float wt[(2 * L2_RMAX + 1) * (2 * L2_RMAX + 1)];
    constexpr bool lds = LDS;
    const int side = 2 * rad + 1, span = L2_T + 2 * rad;
    if (__syncthreads_or(nz) && lds) {
        for (int i = tid; i < span * span; i += L2_T * L2_T) {
            const int yy = i / span, xx = i - yy * span;
            const int hh = ty * L2_T - rad + yy, ww = tx * L2_T - rad + xx;
            const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const int p = hh * W + ww;
            tg[0][yy * L2_S + xx] = in ? Tg[p] : 0.f; tg[1][yy * L2_S + xx] = in ? Tg[P + p] : 0.f; tg[2][yy * L2_S + xx] = in ? Tg[2 * P + p] : 0.f;
        }
        for (int i = tid; i < side * side; i += L2_T * L2_T) {
            const int dh = i / side - rad, dw = i % side - rad;
            wt[i] = fmaxf(diam - sqrtf((float)(dh * dh) + (float)(dw * dw)), 0.f);                     // :224-225
        }
        __syncthreads();
    }
    float lsum = 0.f, cnt = 0.f, any = 0.f;
    if (inside) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (nz) {
            any = ((h | w) != 0) ? 1.f : 0.f;                           // `if rendering_nonzero_idxs.sum()` (:214)
            // every pixel outside the window has weight 0: masked target 0, distance ||r||  (:223-231)
            float best = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
            float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll UNR
            for (int dh = -rad; dh <= rad; ++dh) {
                const int hh = h + dh;
                if (hh < 0 || hh >= H) continue;
#pragma unroll UNR
                for (int dw = -rad; dw <= rad; ++dw) {
                    const int ww = w + dw;
                    if (ww < 0 || ww >= W) continue;
                    float wgt, t0, t1, t2;
                    if (lds) {
                        const int o = (ly + dh + rad) * L2_S + (lx + dw + rad);
                        wgt = wt[(dh + rad) * side + (dw + rad)];
                        t0 = tg[0][o]; t1 = tg[1][o]; t2 = tg[2][o];
                    } else {
                        const int p = hh * W + ww;
                        wgt = fmaxf(diam - sqrtf((float)(dh * dh) + (float)(dw * dw)), 0.f);
                        t0 = Tg[p]; t1 = Tg[P + p]; t2 = Tg[2 * P + p];
                    }
                    const float v0 = t0 * wgt, v1 = t1 * wgt, v2 = t2 * wgt;                           // :227
                    const float e0 = v0 - r0, e1 = v1 - r1, e2 = v2 - r2;
                    const float d = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);                                // :232
                    if (d < best) { best = d; b0 = v0; b1 = v1; b2 = v2; }
                }
            }
            if (best < threshold_nocs) {                                 // :234
                lsum = best;
                cnt = 1.f;
                if (best > 0.f) { g0 = (r0 - b0) / best; g1 = (r1 - b1) / best; g2 = (r2 - b2) / best; }
            }
        }
        G[q] = g0; G[P + q] = g1; G[2 * P + q] = g2;
    }
    __shared__ float red[3][4];
    float v[3] = {lsum, cnt, any};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float x = v[i];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if ((tid & 63) == 0) red[i][tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 3) partial[((int64_t)b * nblk + blk) * 3 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

template <bool LDS, int RADC>
__global__ __launch_bounds__(L2_T * L2_T) void sdfr_loss_2d_pixels_kernel(const float* __restrict__ rend, const float* __restrict__ target,
                                                                         int H, int W, const int32_t* __restrict__ wh, int pst, float diam,
                                                                         float threshold_nocs, float* __restrict__ g_rend,
                                                                         float* __restrict__ partial) {
    loss_2d_pixels_block<LDS, RADC>(blockIdx.x, blockIdx.y, gridDim.x, rend, target, H, W, wh, pst, diam, threshold_nocs, g_rend, partial);
}

__global__ __launch_bounds__(256) void sdfr_loss_2d_finalize_kernel(const float* __restrict__ partial, int nblk, int P, float weight,
                                                                   float* __restrict__ loss, float* __restrict__ g_rend,
                                                                   int32_t* __restrict__ nvalid) {
    const int b = blockIdx.y, tid = threadIdx.x;
    __shared__ float red[3][256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = tid; i < nblk; i += 256) {
        const float* p = partial + ((int64_t)b * nblk + i) * 3;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; red[2][tid] += red[2][tid + st]; }
        __syncthreads();
    }
    const float tot = red[0][0], cf = red[1][0], anyf = red[2][0];
    const float inv = cf > 0.f ? 1.f / cf : 0.f;
    const float k = (anyf > 0.f) ? weight * inv : 0.f;                    // mean over the selected pixels (:234), times the loss weight
    const int64_t e = (int64_t)blockIdx.x * 256 + tid;
    if (e < (int64_t)3 * P) g_rend[(int64_t)b * 3 * P + e] *= k;
    if (blockIdx.x == 0 && tid == 0) {
        // no non-zero pixel -> 0 (:235-236); non-zero pixels but none under the threshold -> mean of an empty set = NaN (:234)
        loss[b] = (anyf > 0.f) ? (cf > 0.f ? tot * inv : __int_as_float(0x7fc00000)) : 0.f;
        nvalid[b] = (int)cf;
    }
}

static int loss_2d_impl(const char* who, const float* rend, const float* target, int B, int H, int W, const int32_t* wh, int pst, int nblk,
                        float diam, float threshold_nocs, float weight, float* loss, float* g_rend, int32_t* nvalid, float* scratch, void* stream) {
    SDFR_REQUIRE(rend && target && loss && g_rend && nvalid && scratch, "%s: NULL argument", who);
    SDFR_REQUIRE(H > 0 && W > 0 && diam > 0.f && pst > 0 && nblk > 0, "%s: bad size", who);
    if (B <= 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int rad = (int)ceilf(diam) - 1;
    if (rad == 4)                                                        // the loop's diam = 5 (optimizer.py:200)
        hipLaunchKernelGGL((sdfr_loss_2d_pixels_kernel<true, 4>), dim3(nblk, B), dim3(L2_T * L2_T), 0, s, rend, target, H, W, wh, pst, diam,
                           threshold_nocs, g_rend, scratch);
    else if (rad <= L2_RMAX)
        hipLaunchKernelGGL((sdfr_loss_2d_pixels_kernel<true, 0>), dim3(nblk, B), dim3(L2_T * L2_T), 0, s, rend, target, H, W, wh, pst, diam,
                           threshold_nocs, g_rend, scratch);
    else
        hipLaunchKernelGGL((sdfr_loss_2d_pixels_kernel<false, 0>), dim3(nblk, B), dim3(L2_T * L2_T), 0, s, rend, target, H, W, wh, pst, diam,
                           threshold_nocs, g_rend, scratch);
    SDFR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sdfr_loss_2d_finalize_kernel, dim3(sdfr_cdiv(3 * pst, 256), B), dim3(256), 0, s, scratch, nblk, pst, weight, loss, g_rend,
                       nvalid);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_loss_2d(const float* rend, const float* target, int B, int H, int W, float diam, float threshold_nocs, float weight,
                            float* loss, float* g_rend, int32_t* nvalid, float* scratch, void* stream) {
    SDFR_REQUIRE(H > 0 && W > 0, "sdfr_loss_2d: bad size");
    return loss_2d_impl("sdfr_loss_2d", rend, target, B, H, W, nullptr, H * W, sdfr_cdiv(W, L2_T) * sdfr_cdiv(H, L2_T), diam, threshold_nocs,
                        weight, loss, g_rend, nvalid, scratch, stream);
}

// ragged extents: crop b compares its own W_b x H_b image (wh int32[B][2] on the device); rend / target / g_rend in slots of pix_stride
// pixels per channel ([B][3][pix_stride]); tiles16_cap >= ceil(W_b/16) ceil(H_b/16) for every crop; scratch float[3 * B * tiles16_cap].
extern "C" int sdfr_loss_2d_r(const float* rend, const float* target, int B, const int32_t* wh, int pix_stride, int tiles16_cap, float diam,
                              float threshold_nocs, float weight, float* loss, float* g_rend, int32_t* nvalid, float* scratch, void* stream) {
    SDFR_REQUIRE(wh, "sdfr_loss_2d_r: NULL extents");
    return loss_2d_impl("sdfr_loss_2d_r", rend, target, B, 1, 1, wh, pix_stride, tiles16_cap, diam, threshold_nocs, weight, loss, g_rend, nvalid,
                        scratch, stream);
}

// ---- both losses in TWO launches instead of four (r06) ----------------------------------------------------------------------------------
// The refinement loop evaluates the two losses on the same rendering and they do not depend on each other: blocks [0, nblk2) of a crop run the
// 2-D pixel pass, blocks [nblk2, nblk2 + nblk3) the 3-D pairs pass (both 256 threads) -- one launch.  A second, tiny launch (two blocks per
// crop) re-reduces the partials in the finalize kernels' fixed order -- loss, nvalid / npairs and g_scale carry the same bits -- and, instead
// of rescaling the gradient arrays in a sweep of 3 P + 3 cap floats per crop, publishes the factor: kscale[b] = (weight_2d / n_valid or 0,
// weight_3d / n_pairs or 0).  The consumers multiply on load (sdfr_splat_backward_x: g_color * k2; sdfr_pose_latent_solver: g_xyzf * k3) --
// the product the finalize kernels stored.
// (A first version let the LAST block of each kind do the finalize behind a ticket counter: one launch, but the device-scope fence every block
// needs before its ticket writes the XCD's L2 back -- the L2s are not coherent with each other -- and 24 000 blocks of a 64-crop launch spent
// 2.3 ms doing so, where the four r05 launches took 0.22 ms.  profiles/r06_notes.md.)
struct LossesArgs {
    // 2-D
    const float* rend; const float* target; int H, W; const int32_t* wh; int pst; float diam, threshold_nocs, w2;
    float* loss2d; float* g_rend; int32_t* nvalid; float* part2; int nblk2;
    // 3-D
    const float* est; const int32_t* ecnt; int ecap; const float* lidar; const int32_t* lcnt; int lcap; const float* scale; float threshold3, w3;
    float* loss3d; float* g_est; float* g_scale; int32_t* npairs; float* part3; int nblk3;
    float* kscale;
};

template <bool LDS, int RADC>
__global__ __launch_bounds__(256) void sdfr_losses_fused_kernel(const LossesArgs A) {
    const int b = blockIdx.y;
    if ((int)blockIdx.x < A.nblk2)
        loss_2d_pixels_block<LDS, RADC>(blockIdx.x, b, A.nblk2, A.rend, A.target, A.H, A.W, A.wh, A.pst, A.diam, A.threshold_nocs, A.g_rend, A.part2);
    else
        loss_3d_pairs_block(blockIdx.x - A.nblk2, b, A.nblk3, A.est, A.ecnt, A.ecap, A.lidar, A.lcnt, A.lcap, A.scale, A.threshold3, A.g_est, A.part3);
}

// the fixed-order re-reduction of the finalize kernels (same statements, same order)
__device__ __forceinline__ void reduce3_fixed(const float* __restrict__ partial, int nblk, int b, float (*red)[256], float& r0, float& r1, float& r2) {
    const int tid = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = tid; i < nblk; i += 256) {
        const float* p = partial + ((int64_t)b * nblk + i) * 3;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; red[2][tid] += red[2][tid + st]; }
        __syncthreads();
    }
    r0 = red[0][0]; r1 = red[1][0]; r2 = red[2][0];
}

__global__ __launch_bounds__(256) void sdfr_losses_finalize_kernel(const LossesArgs A) {
    const int b = blockIdx.y, tid = threadIdx.x;
    __shared__ float red[3][256];
    if (blockIdx.x == 0) {
        float tot, cf, anyf;
        reduce3_fixed(A.part2, A.nblk2, b, red, tot, cf, anyf);
        if (tid == 0) {
            const float inv = cf > 0.f ? 1.f / cf : 0.f;
            A.kscale[2 * b] = (anyf > 0.f) ? A.w2 * inv : 0.f;
            A.loss2d[b] = (anyf > 0.f) ? (cf > 0.f ? tot * inv : __int_as_float(0x7fc00000)) : 0.f;
            A.nvalid[b] = (int)cf;
        }
    } else {
        float tot, gst, cf;
        reduce3_fixed(A.part3, A.nblk3, b, red, tot, gst, cf);
        if (tid == 0) {
            const float inv = cf > 0.f ? 1.f / cf : 0.f;
            const int ne = sdfr_count(A.ecnt, b, A.ecap), nl = sdfr_count(A.lcnt, b, A.lcap);
            A.kscale[2 * b + 1] = A.w3 * inv;
            A.loss3d[b] = cf > 0.f ? tot * inv : 0.f;
            A.g_scale[b] = A.w3 * gst * inv;
            A.npairs[b] = (ne > 0 && nl > 0) ? (int)cf : -1;
        }
    }
}

// rend / target / g_rend: [B][3][H*W] (wh == NULL) or ragged slots [B][3][pix_stride] with wh int32[B][2] and tiles16_cap tile slots per crop.
// g_rend and g_est receive the UN-normalised gradients; kscale float[B][2] the factors; scratch2 float[3 * B * tiles], scratch3
// float[3 * B * ceil(ecap / 64)].
extern "C" int sdfr_losses_fused(const float* rend, const float* target, int B, int H, int W, const int32_t* wh, int pix_stride, int tiles16_cap,
                                 float diam, float threshold_nocs, float weight2d, float* loss2d, float* g_rend, int32_t* nvalid, float* scratch2,
                                 const float* est, const int32_t* ecnt, int ecap, const float* lidar, const int32_t* lcnt, int lcap,
                                 const float* scale, float threshold3d, float weight3d, float* loss3d, float* g_est, float* g_scale,
                                 int32_t* npairs, float* scratch3, float* kscale, void* stream) {
    SDFR_REQUIRE(rend && target && loss2d && g_rend && nvalid && scratch2 && est && lidar && scale && loss3d && g_est && g_scale && npairs &&
                 scratch3 && kscale, "sdfr_losses_fused: NULL argument");
    SDFR_REQUIRE(diam > 0.f && ecap > 0 && lcap >= 0, "sdfr_losses_fused: bad size");
    if (B <= 0) return SDFR_OK;
    LossesArgs A;
    A.rend = rend; A.target = target; A.wh = wh; A.diam = diam; A.threshold_nocs = threshold_nocs; A.w2 = weight2d;
    if (wh) { SDFR_REQUIRE(pix_stride > 0 && tiles16_cap > 0, "sdfr_losses_fused: ragged extents need pix_stride and tiles16_cap"); A.H = 1; A.W = 1; A.pst = pix_stride; A.nblk2 = tiles16_cap; }
    else { SDFR_REQUIRE(H > 0 && W > 0, "sdfr_losses_fused: bad image size"); A.H = H; A.W = W; A.pst = H * W; A.nblk2 = sdfr_cdiv(W, L2_T) * sdfr_cdiv(H, L2_T); }
    A.loss2d = loss2d; A.g_rend = g_rend; A.nvalid = nvalid; A.part2 = scratch2;
    A.est = est; A.ecnt = ecnt; A.ecap = ecap; A.lidar = lidar; A.lcnt = lcnt; A.lcap = lcap; A.scale = scale; A.threshold3 = threshold3d; A.w3 = weight3d;
    A.loss3d = loss3d; A.g_est = g_est; A.g_scale = g_scale; A.npairs = npairs; A.part3 = scratch3; A.nblk3 = sdfr_cdiv(ecap, L3_PTS);
    A.kscale = kscale;
    const dim3 grid(A.nblk2 + A.nblk3, B);
    hipStream_t s = (hipStream_t)stream;
    const int rad = (int)ceilf(diam) - 1;
    if (rad == 4) hipLaunchKernelGGL((sdfr_losses_fused_kernel<true, 4>), grid, dim3(256), 0, s, A);
    else if (rad <= L2_RMAX) hipLaunchKernelGGL((sdfr_losses_fused_kernel<true, 0>), grid, dim3(256), 0, s, A);
    else hipLaunchKernelGGL((sdfr_losses_fused_kernel<false, 0>), grid, dim3(256), 0, s, A);
    SDFR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sdfr_losses_finalize_kernel, dim3(2, B), dim3(256), 0, s, A);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- solver step -------------------------------------------------------------------------------------------------------
// params / grads: one flat structure-of-arrays buffer  [ yaw(B) | trans(B,3) | scale(B) | latent(B,L) ]  so that each section is the
// dense array the renderer kernels read; Adam state m, v [B][4], step counter t [B].
__global__ __launch_bounds__(64) void sdfr_solver_step_kernel(float* __restrict__ params, const float* __restrict__ grads, int L,
                                                             const float* __restrict__ loss2d, const float* __restrict__ loss3d,
                                                             const int32_t* __restrict__ npairs, float w2, float w3,
                                                             float* __restrict__ adam_m, float* __restrict__ adam_v,
                                                             int32_t* __restrict__ adam_t, float lr_adam, float lr_scale, float lr_latent,
                                                             int B, float* __restrict__ total, int32_t* __restrict__ stepped) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    sdfr_solver_crop(b, B, params, grads, L, loss2d, loss3d, npairs, w2, w3, adam_m, adam_v, adam_t, lr_adam, lr_scale, lr_latent, total, stepped);
}

extern "C" int sdfr_solver_step(float* params, const float* grads, int L, const float* loss2d, const float* loss3d, const int32_t* npairs,
                                float w2, float w3, float* adam_m, float* adam_v, int32_t* adam_t, float lr_adam, float lr_scale,
                                float lr_latent, int B, float* total, int32_t* stepped, void* stream) {
    SDFR_REQUIRE(params && grads && loss2d && loss3d && npairs && adam_m && adam_v && adam_t && total && stepped,
                 "sdfr_solver_step: NULL argument");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_solver_step_kernel, dim3(sdfr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, params, grads, L, loss2d, loss3d,
                       npairs, w2, w3, adam_m, adam_v, adam_t, lr_adam, lr_scale, lr_latent, B, total, stepped);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
