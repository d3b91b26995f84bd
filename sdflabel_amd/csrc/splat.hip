// Surfel splatting with per-pixel ray/tangent-plane intersection and depth-softmax compositing (gfx950).
//
// Replaces inside_surfel(diam, softclamp=False, add_bg=False) (reference sdfrenderer/renderer/primitives.py:165-242) and the
// compositing of Rasterer.forward (sdfrenderer/renderer/rasterer.py:113-144).  The reference materialises ~10 dense N x P
// tensors although only ~0.04 % of the (surfel, pixel) pairs are covered; here nothing of size N x P ever exists:
//
//   forward   one wavefront per 8x8 pixel tile (lane = pixel).  The wave scans the surfels' conservative screen boxes
//             64 at a time and compacts the overlapping ones with a ballot into an LDS candidate list (ascending surfel
//             order, deterministic).  Candidates are staged 64 at a time into LDS (lane = candidate) and every lane walks
//             them with broadcast LDS reads: pass 1 per-pixel norm nu (:228), pass 2 max logit, pass 3 softmax sums and the
//             composited colour / mask / depth / normals.  Per-pixel softmax state goes to `aux` for the backward.
//   backward  one wavefront per SURFEL (lanes = pixels of its screen box).  Each lane re-evaluates the coverage test with
//             the identical arithmetic, rebuilds its softmax weight from `aux`, and accumulates the surfel's gradients in
//             registers; one wave reduction, no atomics, deterministic.
//
// Autograd semantics reproduced: coverage mask and nu are constants (:226,:228); |n.ray| < 0.01 is overwritten by eps in place
// and passes no gradient through b (:210); clamp(min=0) and clamp(max=1) pass gradient on the closed side.
// Compiled with -ffp-contract=off so that products and sums round separately like the reference's ATen ops.
#include "sdfr_common.h"
#include <float.h>

#define SPL_LC 1024            // LDS candidate-list capacity per tile (beyond it the tile walks every surfel)

struct Hit {
    bool m;        // inside the disc
    bool small;    // |n.ray| < 0.01  (b replaced by eps)
    float t;       // ray parameter of the plane hit (z in the reference, primitives.py:211)
    float b;       // n.ray after the eps substitution
};

// primitives.py:209-226 for one (surfel, pixel) pair
__device__ __forceinline__ Hit splat_eval(float px, float py, float pz, float nx, float ny, float nz, float a, float rx,
                                          float ry, float rz, float diam) {
    Hit h;
    const float b0 = rx * nx + ry * ny + rz * nz;                                // :209
    h.small = fabsf(b0) < 0.01f;                                                 // :210
    h.b = h.small ? FLT_EPSILON : b0;
    h.t = a / h.b;                                                               // :211
    const float vx = px - rx * h.t, vy = py - ry * h.t, vz = pz - rz * h.t;     // :212,:215
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    h.m = (diam - d) > 0.f;                                                      // :220,:226
    return h;
}

__device__ __forceinline__ void pixel_ray(const float* __restrict__ Ki, float x, float y, float& rx, float& ry, float& rz) {
    rx = fmaf(Ki[1], y, Ki[0] * x) + Ki[2];                                      // :203-208
    ry = fmaf(Ki[4], y, Ki[3] * x) + Ki[5];
    rz = fmaf(Ki[7], y, Ki[6] * x) + Ki[8];
}

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
    const float pad = 1.5f + 1e-3f * (fabsf(u1) + fabsf(u2));
    u1 -= pad; u2 += pad;
    if (isnan(u1) || isnan(u2)) return true;                  // undecidable: keep the whole axis
    if (u2 < 0.f || u1 > (float)(n - 1)) return false;       // entirely off screen
    lo = (int)fmaxf(floorf(u1), 0.f);
    hi = (int)fminf(ceilf(u2), (float)(n - 1));
    return lo <= hi;
}

__device__ __forceinline__ bool surfel_bbox(const float* __restrict__ K, float px, float py, float pz, float rho, int W, int H,
                                            int& x0, int& y0, int& x1, int& y1) {
    x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1;
    const bool standard = (K[1] == 0.f) && (K[3] == 0.f) && (K[6] == 0.f) && (K[7] == 0.f) && (K[8] == 1.f);
    if (!standard) return true;
    if (!axis_range(px, pz, rho, K[0], K[2], W, x0, x1)) return false;
    if (!axis_range(py, pz, rho, K[4], K[5], H, y0, y1)) return false;
    return true;
}

__global__ __launch_bounds__(256) void sdfr_splat_bbox_kernel(const float* __restrict__ K, const float* __restrict__ p_cam,
                                                             int cap, const int32_t* __restrict__ cnt, int W, int H, float diam,
                                                             int4* __restrict__ bbox) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= sdfr_count(cnt, b, cap)) return;
    const int64_t e = (int64_t)b * cap + s;
    int x0, y0, x1, y1;
    const bool ok = surfel_bbox(K + (int64_t)b * 9, p_cam[e * 3], p_cam[e * 3 + 1], p_cam[e * 3 + 2], diam, W, H, x0, y0, x1, y1);
    bbox[e] = ok ? make_int4(x0, y0, x1, y1) : make_int4(1, 1, 0, 0);
}

// ---- forward ----------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void sdfr_splat_fwd_kernel(const float* __restrict__ Kinv, const float* __restrict__ p_cam,
                                                           const float* __restrict__ n_cam, const float* __restrict__ attr,
                                                           const int4* __restrict__ bbox, int cap, const int32_t* __restrict__ cnt,
                                                           int W, int H, float diam, float depth_constant,
                                                           float* __restrict__ color, float* __restrict__ mask,
                                                           float* __restrict__ depth, float* __restrict__ normals,
                                                           float* __restrict__ aux) {
    const int b = blockIdx.y;
    const int tilesX = (W + 7) >> 3;
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    const int lane = threadIdx.x;
    const int X0 = tx * 8, Y0 = ty * 8;
    const int X1 = min(X0 + 7, W - 1), Y1 = min(Y0 + 7, H - 1);
    const int x = X0 + (lane & 7), y = Y0 + (lane >> 3);
    const bool inside = (x < W) && (y < H);
    const int count = sdfr_count(cnt, b, cap);
    const int64_t sb = (int64_t)b * cap;

    __shared__ int list[SPL_LC];
    __shared__ float sd[10][64];

    // candidate list: surfels whose conservative box overlaps this tile, ascending order
    int nc = 0;
    for (int s0 = 0; s0 < count; s0 += 64) {
        const int s = s0 + lane;
        bool ov = false;
        if (s < count) {
            const int4 bb = bbox[sb + s];
            ov = !(bb.x > X1 || bb.z < X0 || bb.y > Y1 || bb.w < Y0);
        }
        const unsigned long long bal = __ballot(ov);
        if (ov) {
            const int pos = nc + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < SPL_LC) list[pos] = s;
        }
        nc += __popcll(bal);
    }
    const bool overflow = nc > SPL_LC;
    const int total = overflow ? count : nc;
    __syncthreads();

    float rx, ry, rz;
    pixel_ray(Kinv + (int64_t)b * 9, (float)x, (float)y, rx, ry, rz);

    // walk all candidates: stage 64 at a time into LDS (lane = candidate), then broadcast-read them
    auto for_each = [&](auto&& body) {
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int c = c0 + lane;
            __syncthreads();
            if (c < total) {
                const int s = overflow ? c : list[c];
                const int64_t e = (sb + s) * 3;
                const float px = p_cam[e], py = p_cam[e + 1], pz = p_cam[e + 2];
                const float nx = n_cam[e], ny = n_cam[e + 1], nz = n_cam[e + 2];
                sd[0][lane] = px; sd[1][lane] = py; sd[2][lane] = pz;
                sd[3][lane] = nx; sd[4][lane] = ny; sd[5][lane] = nz;
                sd[6][lane] = nx * px + ny * py + nz * pz;                      // :202
                sd[7][lane] = attr[e]; sd[8][lane] = attr[e + 1]; sd[9][lane] = attr[e + 2];
            }
            __syncthreads();
            const int kn = min(64, total - c0);
            for (int k = 0; k < kn; ++k) body(k);
        }
    };

    // pass 1: nu = || -t * mask ||_2 over the surfels  (:227-228)
    float nu2 = 0.f;
    for_each([&](int k) {
        const Hit h = splat_eval(sd[0][k], sd[1][k], sd[2][k], sd[3][k], sd[4][k], sd[5][k], sd[6][k], rx, ry, rz, diam);
        if (h.m) nu2 += h.t * h.t;
    });
    const float nu = sqrtf(nu2);
    const float nue = nu + FLT_EPSILON;
    // pass 2: max logit over the covering surfels (:229-230,:240)
    float lmax = -FLT_MAX;
    for_each([&](int k) {
        const Hit h = splat_eval(sd[0][k], sd[1][k], sd[2][k], sd[3][k], sd[4][k], sd[5][k], sd[6][k], rx, ry, rz, diam);
        if (h.m) {
            const float q = (-h.t) / nue + 1.f;
            lmax = fmaxf(lmax, fmaxf(q, 0.f) * depth_constant);
        }
    });
    // pass 3: softmax sums and composites (:240, rasterer.py:119-144)
    float den = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dz = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    for_each([&](int k) {
        const Hit h = splat_eval(sd[0][k], sd[1][k], sd[2][k], sd[3][k], sd[4][k], sd[5][k], sd[6][k], rx, ry, rz, diam);
        if (h.m) {
            const float q = (-h.t) / nue + 1.f;
            const float e = expf(fmaxf(q, 0.f) * depth_constant - lmax);
            den += e;
            c0 += e * sd[7][k]; c1 += e * sd[8][k]; c2 += e * sd[9][k];
            dz += e * sd[2][k];
            n0 += e * ((sd[3][k] + 1.f) / 2.f); n1 += e * ((sd[4][k] + 1.f) / 2.f); n2 += e * ((sd[5][k] + 1.f) / 2.f);
        }
    });
    if (!inside) return;
    const bool cov = den > 0.f;
    const float inv = cov ? 1.f / den : 0.f;
    c0 *= inv; c1 *= inv; c2 *= inv; dz *= inv; n0 *= inv; n1 *= inv; n2 *= inv;
    const float ms = cov ? 1.f : 0.f;
    const int P = W * H;
    const int pix = y * W + x;
    unsigned gates = 0;
    gates |= (c0 <= 1.f) ? 1u : 0u; gates |= (c1 <= 1.f) ? 2u : 0u; gates |= (c2 <= 1.f) ? 4u : 0u;
    gates |= 8u;
    gates |= (n0 <= 1.f) ? 16u : 0u; gates |= (n1 <= 1.f) ? 32u : 0u; gates |= (n2 <= 1.f) ? 64u : 0u;
    if (color) {
        float* o = color + (int64_t)b * 3 * P + pix;
        o[0] = fminf(c0, 1.f); o[P] = fminf(c1, 1.f); o[2 * P] = fminf(c2, 1.f);
    }
    if (mask) mask[(int64_t)b * P + pix] = fminf(ms, 1.f);
    if (depth) depth[(int64_t)b * P + pix] = dz;
    if (normals) {
        float* o = normals + (int64_t)b * 3 * P + pix;
        o[0] = fminf(n0, 1.f); o[P] = fminf(n1, 1.f); o[2 * P] = fminf(n2, 1.f);
    }
    if (aux) {
        float4 a4;
        a4.x = nu; a4.y = cov ? lmax : 0.f; a4.z = den; a4.w = __uint_as_float(gates);
        reinterpret_cast<float4*>(aux)[(int64_t)b * P + pix] = a4;
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void sdfr_splat_bwd_kernel(
    const float* __restrict__ K, const float* __restrict__ Kinv, const float* __restrict__ p_cam, const float* __restrict__ n_cam,
    const float* __restrict__ attr, int cap, const int32_t* __restrict__ cnt, int W, int H, float diam, float depth_constant,
    const float* __restrict__ aux, const float* __restrict__ color, const float* __restrict__ mask, const float* __restrict__ depth,
    const float* __restrict__ normals, const float* __restrict__ g_color, const float* __restrict__ g_mask,
    const float* __restrict__ g_depth, const float* __restrict__ g_normals, float* __restrict__ g_p, float* __restrict__ g_n,
    float* __restrict__ g_attr) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= sdfr_count(cnt, b, cap)) return;
    const int64_t e = ((int64_t)b * cap + s) * 3;
    const float px = p_cam[e], py = p_cam[e + 1], pz = p_cam[e + 2];
    const float nx = n_cam[e], ny = n_cam[e + 1], nz = n_cam[e + 2];
    const float a0 = attr[e], a1 = attr[e + 1], a2 = attr[e + 2];
    const float m0 = (nx + 1.f) / 2.f, m1 = (ny + 1.f) / 2.f, m2 = (nz + 1.f) / 2.f;
    const float a = nx * px + ny * py + nz * pz;
    const float* Ki = Kinv + (int64_t)b * 9;
    const int P = W * H;
    int x0, y0, x1, y1;
    float sC0 = 0.f, sC1 = 0.f, sC2 = 0.f, sN0 = 0.f, sN1 = 0.f, sN2 = 0.f, sZ = 0.f, sA = 0.f, sB0 = 0.f, sB1 = 0.f, sB2 = 0.f;
    if (surfel_bbox(K + (int64_t)b * 9, px, py, pz, diam, W, H, x0, y0, x1, y1)) {
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        for (int i = lane; i < bw * bh; i += 64) {
            const int yy = i / bw;
            const int x = x0 + (i - yy * bw), y = y0 + yy;
            float rx, ry, rz;
            pixel_ray(Ki, (float)x, (float)y, rx, ry, rz);
            const Hit h = splat_eval(px, py, pz, nx, ny, nz, a, rx, ry, rz, diam);
            if (!h.m) continue;
            const int pix = y * W + x;
            const float4 ax = reinterpret_cast<const float4*>(aux)[(int64_t)b * P + pix];
            const float nue = ax.x + FLT_EPSILON;
            const unsigned gates = __float_as_uint(ax.w);
            const float q = (-h.t) / nue + 1.f;
            const float w = expf(fmaxf(q, 0.f) * depth_constant - ax.y) / ax.z;
            // gated upstream gradients and S = sum_j w_j dL/dw_j = <gated grads, composited outputs>
            float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, gm = 0.f, gd = 0.f, gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, S = 0.f;
            if (g_color) {
                const float* g = g_color + (int64_t)b * 3 * P + pix;
                const float* o = color + (int64_t)b * 3 * P + pix;
                gc0 = (gates & 1u) ? g[0] : 0.f; gc1 = (gates & 2u) ? g[P] : 0.f; gc2 = (gates & 4u) ? g[2 * P] : 0.f;
                S += gc0 * o[0] + gc1 * o[P] + gc2 * o[2 * P];
            }
            if (g_mask) { gm = g_mask[(int64_t)b * P + pix]; S += gm * mask[(int64_t)b * P + pix]; }
            if (g_depth) { gd = g_depth[(int64_t)b * P + pix]; S += gd * depth[(int64_t)b * P + pix]; }
            if (g_normals) {
                const float* g = g_normals + (int64_t)b * 3 * P + pix;
                const float* o = normals + (int64_t)b * 3 * P + pix;
                gn0 = (gates & 16u) ? g[0] : 0.f; gn1 = (gates & 32u) ? g[P] : 0.f; gn2 = (gates & 64u) ? g[2 * P] : 0.f;
                S += gn0 * o[0] + gn1 * o[P] + gn2 * o[2 * P];
            }
            const float dLdw = gc0 * a0 + gc1 * a1 + gc2 * a2 + gm + gd * pz + gn0 * m0 + gn1 * m1 + gn2 * m2;
            sC0 += w * gc0; sC1 += w * gc1; sC2 += w * gc2;
            sN0 += w * gn0; sN1 += w * gn1; sN2 += w * gn2;
            sZ += w * gd;
            const float dl = w * (dLdw - S);
            const float dq = (q >= 0.f) ? dl * depth_constant : 0.f;
            const float dt = -(dq / nue);                           // zeta = -t * mask
            sA += dt / h.b;                                         // t = a / b
            if (!h.small) {
                const float db = -dt * h.t / h.b;
                sB0 += db * rx; sB1 += db * ry; sB2 += db * rz;
            }
        }
    }
    sC0 = wave_sum(sC0); sC1 = wave_sum(sC1); sC2 = wave_sum(sC2);
    sN0 = wave_sum(sN0); sN1 = wave_sum(sN1); sN2 = wave_sum(sN2);
    sZ = wave_sum(sZ); sA = wave_sum(sA);
    sB0 = wave_sum(sB0); sB1 = wave_sum(sB1); sB2 = wave_sum(sB2);
    if (lane == 0) {
        g_attr[e] = sC0; g_attr[e + 1] = sC1; g_attr[e + 2] = sC2;
        g_n[e] = 0.5f * sN0 + sB0 + sA * px;
        g_n[e + 1] = 0.5f * sN1 + sB1 + sA * py;
        g_n[e + 2] = 0.5f * sN2 + sB2 + sA * pz;
        g_p[e] = sA * nx;
        g_p[e + 1] = sA * ny;
        g_p[e + 2] = sA * nz + sZ;
    }
}

// ---- C ABI --------------------------------------------------------------------------------------------------------

extern "C" int sdfr_splat_forward(const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr,
                                  int B, int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant,
                                  int32_t* bbox_ws, float* color, float* mask, float* depth, float* normals, float* aux,
                                  void* stream) {
    SDFR_REQUIRE(K && Kinv && (cap == 0 || (p_cam && n_cam && attr && bbox_ws)), "sdfr_splat_forward: NULL argument");
    SDFR_REQUIRE(W > 0 && H > 0 && B >= 0 && cap >= 0, "sdfr_splat_forward: bad size");
    if (B == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    if (cap > 0) {
        hipLaunchKernelGGL(sdfr_splat_bbox_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, s, K, p_cam, cap, cnt, W, H, diam,
                           reinterpret_cast<int4*>(bbox_ws));
        SDFR_LAUNCH_CHECK();
    }
    const int tiles = ((W + 7) / 8) * ((H + 7) / 8);
    hipLaunchKernelGGL(sdfr_splat_fwd_kernel, dim3(tiles, B), dim3(64), 0, s, Kinv, p_cam, n_cam, attr,
                       reinterpret_cast<const int4*>(bbox_ws), cap, cnt, W, H, diam, depth_constant, color, mask, depth, normals, aux);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_splat_backward(const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr,
                                   int B, int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant,
                                   const float* aux, const float* color, const float* mask, const float* depth, const float* normals,
                                   const float* g_color, const float* g_mask, const float* g_depth, const float* g_normals,
                                   float* g_p_cam, float* g_n_cam, float* g_attr, void* stream) {
    SDFR_REQUIRE(K && Kinv && aux && g_p_cam && g_n_cam && g_attr, "sdfr_splat_backward: NULL argument");
    SDFR_REQUIRE((!g_color || color) && (!g_mask || mask) && (!g_depth || depth) && (!g_normals || normals),
                 "sdfr_splat_backward: an image gradient was given without the forward image");
    if (B == 0 || cap == 0) return SDFR_OK;
    SDFR_REQUIRE(p_cam && n_cam && attr, "sdfr_splat_backward: NULL surfel array");
    hipLaunchKernelGGL(sdfr_splat_bwd_kernel, dim3(sdfr_cdiv(cap, 4), B), dim3(256), 0, (hipStream_t)stream, K, Kinv, p_cam, n_cam,
                       attr, cap, cnt, W, H, diam, depth_constant, aux, color, mask, depth, normals, g_color, g_mask, g_depth,
                       g_normals, g_p_cam, g_n_cam, g_attr);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
