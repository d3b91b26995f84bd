// Surfel splatting with per-pixel coverage tests and depth-softmax compositing (gfx950).
//
// Replaces the three primitives of the reference renderer and the compositing of Rasterer.forward
// (sdfrenderer/renderer/rasterer.py:92-144) without ever forming the N x P tensors the reference materialises:
//   PRIM 0  'disc'        inside_surfel(diam=0.04, softclamp=False)  primitives.py:165-242   -- the optimizer's primitive:
//                         per-pixel ray / tangent-plane intersection, 3-D disc test, logits normalised by a per-pixel norm
//   PRIM 1  'circle'      inside_circle(diam=0.02, softclamp=True)   primitives.py:4-71      -- 2-D circle; the reference thresholds
//                         sigmoid(.) > 0 (:55), i.e. coverage ends where exp overflows (~29.6 px beyond the circle), and its
//                         softmax runs over z*mask (:70): uncovered surfels keep logit 0 in the denominator
//   PRIM 2  'circle_opt'  inside_circle_opt(diam=0.025)              primitives.py:74-162    -- every surfel stamps the 15x15 pixel
//                         square at trunc(uv + offset), indices clamped into the image (:122-127)
// plus the optional background row (add_bg; primitives.py:64-67,146-153,233-237; rasterer.py:107-111).
//
//   forward   one wavefront per 8x8 pixel tile (lane = pixel).  The wave scans the surfels' conservative screen boxes 64 at a time and
//             compacts the overlapping ones with a ballot into an LDS candidate list (ascending surfel order, deterministic).
//             Candidates are staged 64 at a time into LDS (lane = candidate) and every lane walks them with broadcast LDS reads:
//             [disc: per-pixel norm nu (:228)], max logit, softmax sums + composited colour / mask / depth / normals.
//             Per-pixel softmax state goes to `aux` for the backward.
//   backward  one wavefront per SURFEL.  The lanes re-evaluate the coverage test over its screen box with the identical arithmetic and queue
//             the covered pixels (r05: ballot-compacted in LDS, so that the gradient chain runs on full lanes); each queued pixel rebuilds its
//             softmax weight from `aux` and accumulates the surfel's gradients in registers; one wave reduction, no atomics, bit-repeatable.
//
// Autograd semantics reproduced: coverage masks and norms are constants (:55,:59 / :155,:142 / :226,:228); for the disc |n.ray| < 0.01 is
// overwritten by eps in place and passes no gradient through b (:210); clamp(min=0) / clamp(max=1) pass gradient on the closed side.
// The background logit is treated as a constant (its weight is exactly 0 or 1 for disc and circle_opt; the circle's is handled by the
// host layer).  Compiled with -ffp-contract=off so that products and sums round separately like the reference's ATen ops.
#include "sdfr_common.h"
#include "splat_bbox.h"
#include <float.h>
#include <stdlib.h>
#include <math.h>

// LDS candidate-list capacity per tile (beyond it the tile walks EVERY surfel of the crop, twice, without coverage ballots).  3072 since r06 (6 KiB of 16-bit slots: the one-wave-per-tile geometry keeps its 16 tiles per CU): at the
// reference's shipped rendering_area 32 (configs/config_refine.ini:13) a crop is ~18 tiles and its densest tile lists 1200-1600 of the crop's ~2700
// surfels; with 1024 slots that tile took the walk-everything path and WAS the launch (142 us at one crop, 136 us at 16: tools/splat32_diag.py)
#define SPL_LC 3072
#define SPL_SORT_MAX 1024      // binned tiles with more entries build their list by scanning the boxes (same list)
#define SPL_LCOV 2048          // coverage ballots kept per tile by the wide geometry (16 KiB)
#define SPL_BQ 128             // backward: covered-pixel queue per wave (drained whenever fewer than 64 slots are free)
#define SIGMOID_REACH 29.65f   // (r - d) * 3 > -88.73  <=>  d < r + 29.58: conservative reach of inside_circle's sigmoid(.) > 0

struct SplatArgs {
    const float* K; const float* Kinv;
    const float* p_cam; const float* n_cam; const float* attr;
    const float* uv; const float* znorm;          // PRIM 1,2: clamped 2-D projections [B][cap][2]; || depths ||_2 per crop [B]
    const float* bg; const float* bg_logit;       // optional background image [B][3][H][W] and its logit [B]
    int cap; const int32_t* cnt; int W, H;
    float diam, depth_constant;
    float cover_sq;                               // PRIM 0: coverage threshold on the squared distance (disc_cover_sq)
    // clamp configuration of the primitive (r05).  clamp_c: the softclamp_constant of the sigmoid clamps (3 for the renderer's circle,
    // primitives.py:13; reach = how far beyond the radius sigmoid((r - d) c) stays positive in float32).  Kernels instantiated with ALT take
    // the OTHER clamp of their primitive, which Rasterer.forward never passes but the standalone functions offer: disc -> softclamp=True
    // (sigmoid, :217-218: positive until exp overflows, i.e. practically every pixel), circle / circle_opt -> softclamp=False (hard edge, :51-53,:120)
    float clamp_c, reach;
    // ragged extents (r04): wh != NULL -> crop b renders W_b = wh[2b] x H_b = wh[2b+1] pixels (every crop of a batch its own image size, as
    // the reference pipeline's crops have: utils/refinement.py:586-609); its images live in slots of `pst` pixels per channel
    // ([B][C][pst], rows of W_b pixels in the first W_b H_b entries).  wh == NULL: every crop W x H, pst = W H (the dense layout).
    const int32_t* wh; int pst;
    int64_t bin_stride;                           // words per crop of the tile-list workspace (splat_bbox.h)
};

// image extents of crop b and the pixel stride of its image channels
__device__ __forceinline__ void splat_dims(const SplatArgs& A, int b, int& W, int& H, int& PS) {
    W = A.W; H = A.H; PS = W * H;
    if (A.wh) {
        W = A.wh[2 * b]; H = A.wh[2 * b + 1]; PS = A.pst;
        // an extent outside the caller's contract (include/sdfr.h: 1 <= W_b, H_b and W_b H_b <= pix_stride) renders as an EMPTY crop instead of
        // writing out of bounds (the Python layer validates extents before they reach the device)
        if (W < 1 || H < 1 || (int64_t)W * H > (int64_t)PS || W > 65535 || H > 65535) { W = 0; H = 0; }     // (16-bit pixel coordinates: the backward's queue)
    }
}

struct Hit {
    bool m;        // covered
    bool small;    // disc: |n.ray| < 0.01  (b replaced by eps)
    float t;       // disc: ray parameter of the plane hit (z in the reference, primitives.py:211)
    float b;       // disc: n.ray after the eps substitution
};

// primitives.py:209-226 for one (surfel, pixel) pair.  The coverage test  diam - ||v|| > 0  (:220,:226) is taken on the squared norm:
// with a correctly rounded square root,  sqrt(x) < diam  <=>  x <= cover_sq  for the float cover_sq computed by disc_cover_sq() below --
// the same decisions as the reference's norm-and-compare for every float x, without the square root.
// primitives.py:217-218,:226 (softclamp=True): sigmoid((diam - ||v||) * c) > 0 with torch's float32 sigmoid 1 / (1 + exp(-x))
__device__ __forceinline__ bool soft_disc_cover(float d2, float diam, float c) {
    const float arg = (diam - sqrtf(d2)) * c;
    return (1.f / (1.f + expf(-arg))) > 0.f;
}

template <bool ALT = false>
__device__ __forceinline__ Hit disc_eval(float px, float py, float pz, float nx, float ny, float nz, float a, float rx, float ry, float rz,
                                         float cover_sq, float diam = 0.f, float clamp_c = 0.f) {
    Hit h;
    const float b0 = rx * nx + ry * ny + rz * nz;                                // :209
    h.small = fabsf(b0) < 0.01f;                                                 // :210
    h.b = h.small ? FLT_EPSILON : b0;
    h.t = a / h.b;                                                               // :211
    const float vx = px - rx * h.t, vy = py - ry * h.t, vz = pz - rz * h.t;     // :212,:215
    if (ALT) h.m = soft_disc_cover(vx * vx + vy * vy + vz * vz, diam, clamp_c);
    else h.m = (vx * vx + vy * vy + vz * vz) <= cover_sq;                        // :220,:226
    return h;
}

// largest float x with RN(sqrt(x)) < diam: RN(sqrt(x)) < diam  <=>  sqrt(x) < m, m = midpoint of diam and its predecessor (25 significant
// bits, so m*m is exact in double; sqrt(x) = m would need x = m*m, which is no float) <=>  x < m*m
static float disc_cover_sq(float diam) {
    if (!(diam > 0.f)) return -1.f;                                              // nothing is covered (x >= 0 always)
    const float pred = nextafterf(diam, 0.f);
    const double m = ((double)pred + (double)diam) * 0.5;
    const double m2 = m * m;
    float t = (float)m2;
    if ((double)t >= m2) t = nextafterf(t, 0.f);
    return t;
}

// primitives.py:42-49,55: sigmoid((r - ||uv - pixel||) * softclamp_constant) > 0;  ALT (softclamp=False, :51-53): clamp(r - d, min=0) > 0
template <bool ALT = false>
__device__ __forceinline__ bool circle_cover(float u, float v, float r, float x, float y, float c) {
    const float dx = u - x, dy = v - y;
    const float d = sqrtf(dx * dx + dy * dy);
    if (ALT) return (r - d) > 0.f;
    const float arg = (r - d) * c;
    return (1.f / (1.f + expf(-arg))) > 0.f;
}

// primitives.py:122-127: is pixel coordinate `p` hit by clamp(trunc(u + o), 0, n-1) for some integer offset o in [-7,7] ?
__device__ __forceinline__ bool stamp_axis(float u, int p, int n) {
    const int c = (int)ceilf((float)p - u);
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
        int o = c + d;
        o = o < -7 ? -7 : (o > 7 ? 7 : o);
        int t = (int)truncf(u + (float)o);
        t = t < 0 ? 0 : (t > n - 1 ? n - 1 : t);
        if (t == p) return true;
    }
    // borders collect everything that is clamped onto them
    if (p == 0) { int t = (int)truncf(u - 7.f); if (t <= 0) return true; }
    if (p == n - 1) { int t = (int)truncf(u + 7.f); if (t >= n - 1) return true; }
    return false;
}

// circle_opt with softclamp=False (primitives.py:120): a stamp offset (ox, oy) contributes clamp(r - ||(ox, oy)||, 0), and the sparse tensor
// sums what lands on the same pixel (:135-138), so a pixel is covered iff SOME offset that maps to it lies inside the circle -- iff the one
// with the smallest |ox| and the smallest |oy| does.  Offsets that map to pixel coordinate p on one axis: smallest |o|, or -1 if none.
__device__ __forceinline__ int stamp_axis_min(float u, int p, int n) {
    int best = -1;
    for (int o = -7; o <= 7; ++o) {
        int t = (int)truncf(u + (float)o);
        t = t < 0 ? 0 : (t > n - 1 ? n - 1 : t);
        const int ao = o < 0 ? -o : o;
        if (t == p && (best < 0 || ao < best)) best = ao;
    }
    return best;
}

template <bool ALT = false>
__device__ __forceinline__ bool stamp_cover(float u, float v, int x, int y, int W, int H, float r) {
    if (!ALT) return stamp_axis(u, x, W) && stamp_axis(v, y, H);
    const int ox = stamp_axis_min(u, x, W), oy = stamp_axis_min(v, y, H);
    if (ox < 0 || oy < 0) return false;
    return (r - sqrtf((float)(ox * ox + oy * oy))) > 0.f;
}

// per-surfel depth logit  clamp(-z / (||z|| + eps) + 1, 0) * C   (primitives.py:57-61, :141-144)
__device__ __forceinline__ float depth_logit(float z, float zn, float C, float* q_out) {
    const float q = (-z) / (zn + FLT_EPSILON) + 1.f;
    if (q_out) *q_out = q;
    return fmaxf(q, 0.f) * C;
}

__device__ __forceinline__ void pixel_ray(const float* __restrict__ Ki, float x, float y, float& rx, float& ry, float& rz) {
    rx = fmaf(Ki[1], y, Ki[0] * x) + Ki[2];                                      // :203-208
    ry = fmaf(Ki[4], y, Ki[3] * x) + Ki[5];
    rz = fmaf(Ki[7], y, Ki[6] * x) + Ki[8];
}

__device__ __forceinline__ bool interval(float lo_f, float hi_f, int n, int& lo, int& hi) {
    lo = 0; hi = n - 1;
    if (isnan(lo_f) || isnan(hi_f)) return true;
    if (hi_f < 0.f || lo_f > (float)(n - 1)) return false;
    lo = (int)fmaxf(floorf(lo_f), 0.f);
    hi = (int)fminf(ceilf(hi_f), (float)(n - 1));
    return lo <= hi;
}

template <int PRIM, bool ALT = false>
__device__ __forceinline__ bool surfel_bbox(const SplatArgs& A, int b, int64_t e, int& x0, int& y0, int& x1, int& y1) {
    int W, H, PS_;
    splat_dims(A, b, W, H, PS_);
    x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1;
    const float* K = A.K + (int64_t)b * 9;
    if (PRIM == 0) {
        if (ALT) return true;                                  // the soft clamp covers (practically) every pixel: the whole image
        return disc_bbox(K, A.p_cam[e * 3], A.p_cam[e * 3 + 1], A.p_cam[e * 3 + 2], A.diam, W, H, x0, y0, x1, y1);
    } else if (PRIM == 1) {
        const float u = A.uv[e * 2], v = A.uv[e * 2 + 1];
        const float r = fabsf(K[0] * A.diam / (A.p_cam[e * 3 + 2] + FLT_EPSILON)) + (ALT ? 0.f : A.reach) + 1.f;
        if (!interval(u - r, u + r, W, x0, x1)) return false;
        if (!interval(v - r, v + r, H, y0, y1)) return false;
        return true;
    } else {
        const float u = A.uv[e * 2], v = A.uv[e * 2 + 1];      // clamped stamps always land inside the image
        interval(u - 9.f, u + 9.f, W, x0, x1);
        interval(v - 9.f, v + 9.f, H, y0, y1);
        if (u - 9.f > (float)(W - 1)) { x0 = W - 1; x1 = W - 1; }
        if (u + 9.f < 0.f) { x0 = 0; x1 = 0; }
        if (v - 9.f > (float)(H - 1)) { y0 = H - 1; y1 = H - 1; }
        if (v + 9.f < 0.f) { y0 = 0; y1 = 0; }
        return true;
    }
}

template <int PRIM, bool ALT = false>
__global__ __launch_bounds__(256) void sdfr_splat_bbox_kernel(const SplatArgs A, int4* __restrict__ bbox) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= sdfr_count(A.cnt, b, A.cap)) return;
    const int64_t e = (int64_t)b * A.cap + s;
    int x0, y0, x1, y1;
    const bool ok = surfel_bbox<PRIM, ALT>(A, b, e, x0, y0, x1, y1);
    bbox[e] = ok ? make_int4(x0, y0, x1, y1) : make_int4(1, 1, 0, 0);
}

// screen boxes -> per-tile surfel lists, one workgroup per crop (the drop-in path; the batched path bins inside sdfr_surfels_forward)
__global__ __launch_bounds__(1024) void sdfr_splat_bin_kernel(const SplatArgs A, const int4* __restrict__ bbox, int32_t* __restrict__ bins) {
    __shared__ int tile_cnt[SPL_BIN_MAX_TILES];
    __shared__ int wsum[1024 / 64 + 1];
    const int b = blockIdx.x;
    int W, H, PS_;
    splat_dims(A, b, W, H, PS_);
    sdfr_bin_boxes<1024>(bbox + (int64_t)b * A.cap, sdfr_count(A.cnt, b, A.cap), W, H, A.cap, bins + (int64_t)b * A.bin_stride, tile_cnt, wsum);
}

extern "C" int64_t sdfr_splat_ws_words(int B, int cap, int W, int H) {
    return (int64_t)B * ((int64_t)cap * 4 + sdfr_splat_bin_stride(cap, W, H));
}

// ---- forward ----------------------------------------------------------------------------------------------------

#define SPL_NS 8               // candidate shares per 8x8 pixel tile: the forward is a latency chain over the tile's candidates, split SPL_NS ways
                               // (measured per crop, one wave per share: 1 share 71 us, 4 shares 26 us, 8 shares 21.5 us).  The share
                               // partition defines the summation order of the result, so it is the same in both launch geometries below.

struct SplatAcc {               // online-softmax state of one share for one pixel
    float lmax, cs, c0, c1, c2, dz, n0, n1, n2;
};
__device__ __forceinline__ void acc_rescale(SplatAcc& a, float newmax) {
    const float f = expf(a.lmax - newmax);            // exp(-inf) = 0 on the first hit
    a.cs *= f; a.c0 *= f; a.c1 *= f; a.c2 *= f; a.dz *= f; a.n0 *= f; a.n1 *= f; a.n2 *= f;
    a.lmax = newmax;
}

// One workgroup of PW waves per 8x8 pixel tile; lane = pixel.  (1) The waves scan the crop's surfel boxes together, 64*PW per step,
// and merge their ballots in surfel order into the tile's candidate list (ascending, deterministic) -- or take the tile's entries of
// the crop's tile lists.  (2) Each round of 64*SPL_NS candidates is cut into SPL_NS contiguous shares; a share is staged in LDS
// (lane = candidate) and walked for the 64 pixels with broadcast LDS reads; the per-pixel partial states of the shares (nu^2; then the
// online-softmax running maximum and sums) are merged in share order, so the result is reproducible bit for bit.
//   PW = SPL_NS  one wave per share: the shortest latency chain per tile, 4 tiles per CU -- few crops (a launch that does not fill the chip)
//   PW = 1       one wave walks the shares one after the other (states in registers, no workgroup barrier, 7 KiB of LDS): the same
//                arithmetic in the same order, so the same bits, with 4x the tiles in flight per CU and no idle waves on light tiles --
//                launches of many crops, where the candidate walk is VALU-throughput-bound rather than a latency chain
// disc: the first sweep also records which pixels each candidate covers (one 64-bit ballot per candidate); the second sweep skips
// candidates that cover no pixel of the tile and re-evaluates only the plane hit for the others (PW = 1 keeps ballots for tiles of at most
// 64 candidates and re-evaluates the coverage otherwise: identical arithmetic).
template <int PRIM, int PW, bool ALT = false>
__global__ __launch_bounds__(64 * PW) void sdfr_splat_fwd_kernel(const SplatArgs A, const int4* __restrict__ bbox,
                                                                const int32_t* __restrict__ bins, float* __restrict__ color,
                                                                float* __restrict__ mask, float* __restrict__ depth,
                                                                float* __restrict__ normals, float* __restrict__ aux) {
    static_assert(PW == 1 || PW == SPL_NS, "one wave per share, or one wave for all shares");
    constexpr int SPW = SPL_NS / PW;                   // shares per wave
    constexpr int NT = 64 * PW;
    constexpr int LCOV = (PW == 1) ? 64 : SPL_LCOV;    // coverage ballots kept per tile
    int tile, b;
    sdfr_xcd_crop_map(tile, b);        // a crop's tiles on one XCD: its surfel arrays / tile lists are fetched by one L2
    int W, H, PS;
    splat_dims(A, b, W, H, PS);
    const int tilesX = (W + 7) >> 3;
    if (tile >= tilesX * ((H + 7) >> 3)) return;       // (ragged extents: the launch covers the largest crop's tile count)
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int X0 = tx * 8, Y0 = ty * 8;
    const int X1 = min(X0 + 7, W - 1), Y1 = min(Y0 + 7, H - 1);
    const int x = X0 + (lane & 7), y = Y0 + (lane >> 3);
    const bool inside = (x < W) && (y < H);
    const int count = sdfr_count(A.cnt, b, A.cap);
    const int64_t sb = (int64_t)b * A.cap;
    const float diam = A.diam, C = A.depth_constant;

    // the tile's candidate list: SPL_LC slots of 16 bits (surfel slots < cap <= 65536), or half as many 32-bit ones for larger capacities -- the
    // SAME capacity in both launch geometries, so a tile takes the same path (list / walk-everything) and gives the same bits in either
    __shared__ unsigned short list16[SPL_LC];
    const bool wide_idx = A.cap > 65536;
    const int LCe = wide_idx ? SPL_LC / 2 : SPL_LC;
    auto lset = [&](int i, int v) { if (wide_idx) reinterpret_cast<int*>(list16)[i] = v; else list16[i] = (unsigned short)v; };
    auto lget = [&](int i) -> int { return wide_idx ? reinterpret_cast<const int*>(list16)[i] : (int)list16[i]; };
    __shared__ unsigned long long cov[LCOV];
    __shared__ float sd[PW][11][64];
    __shared__ float nured[PW][64];
    __shared__ int wc[2][PW];
    float (*red)[11][64] = sd;        // the final merge reuses each wave's own staging slice (dead by then); wide geometry: 49 KiB of LDS, 3 tiles per CU

    // ---- (1) candidate list: surfels whose conservative box overlaps this tile, ascending order --------------------------------
    // binned: the tile's entries of the per-crop tile lists (sdfr_bin_boxes: count -> scan -> fill, arbitrary order), sorted here by rank
    // so that every sum below runs in ascending surfel order whatever the fill order was -- a tile touches its candidates, not all N boxes.
    // Otherwise (no lists, or the crop's lists overflowed) the waves scan all boxes together.
    int nc = 0;
    bool binned = false;
    if (bins) {
        const int T = tilesX * ((H + 7) >> 3);
        const int32_t* toff = bins + (int64_t)b * A.bin_stride;
        if (toff[T + 1] == 1) {
            binned = true;
            const int o0 = toff[tile];
            nc = toff[tile + 1] - o0;
            const int32_t* tl = toff + T + 2 + o0;
            // (the rank sort below is quadratic in the tile's entries: a dense tile -- more than SPL_SORT_MAX -- scans the crop's boxes instead,
            // a few steps of ballot compaction that deliver the same ascending list)
            if (nc > SPL_SORT_MAX) { binned = false; nc = 0; }
            else if (PW == 1) {
                if (nc > 0 && nc <= 64) {                  // ranks through lane reads
                    const int v = (lane < nc) ? tl[lane] : 0x7fffffff;
                    int r = 0;
                    for (int j = 0; j < nc; ++j) r += (__builtin_amdgcn_readlane(v, j) < v) ? 1 : 0;
                    if (lane < nc) lset(r, v);
                } else if (nc <= LCe) {
                    for (int i = lane; i < nc; i += 64) {
                        const int v = tl[i];
                        int r = 0;
                        for (int j = 0; j < nc; ++j) r += (tl[j] < v) ? 1 : 0;
                        lset(r, v);
                    }
                }
            } else if (nc > 0 && nc <= LCe) {
                int* tmp = reinterpret_cast<int*>(cov);
                for (int i = threadIdx.x; i < nc; i += NT) tmp[i] = tl[i];
                __syncthreads();
                for (int i = threadIdx.x; i < nc; i += NT) {
                    const int v = tmp[i];
                    int r = 0;
                    for (int j = 0; j < nc; ++j) r += (tmp[j] < v) ? 1 : 0;
                    lset(r, v);
                }
            }
        }
    }
    if (!binned) {
        auto overlaps = [&](int s) {
            if (s >= count) return false;
            const int4 bb = bbox[sb + s];
            // (an EMPTY box -- the off-screen marker (1, 1, 0, 0) -- overlaps nothing: r01-r05 let it pass the test below for tile (0, 0), where
            // it covered no pixel but lengthened the candidate list, i.e. changed the share partition and with it the last bits of that
            // tile's sums relative to the binned path, which skips empty boxes: a crop with off-screen surfels then differed between a
            // launch of 1-3 crops and one of 4+ -- found by optimize_many's bit-identity check, r06)
            if (bb.x > bb.z || bb.y > bb.w) return false;
            return !(bb.x > X1 || bb.z < X0 || bb.y > Y1 || bb.w < Y0);
        };
        bool ov = overlaps(wave * 64 + lane);
        int par = 0;
        for (int s0 = 0; s0 < count; s0 += NT, par ^= 1) {
            const int s = s0 + wave * 64 + lane;
            const bool ovn = overlaps(s + NT);                       // next step's boxes are in flight across the barrier
            const unsigned long long bal = __ballot(ov);
            int off = nc, tot = 0;
            if (PW > 1) {
                if (lane == 0) wc[par][wave] = __popcll(bal);
                __syncthreads();
#pragma unroll
                for (int w = 0; w < PW; ++w) { const int c = wc[par][w]; off += (w < wave) ? c : 0; tot += c; }
            } else {
                tot = __popcll(bal);
            }
            if (ov) {
                const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < LCe) lset(pos, s);
            }
            nc += tot;
            ov = ovn;
        }
    }
    const bool overflow = nc > LCe;
    const int total = overflow ? count : nc;
    const bool use_cov = !overflow && total <= LCOV;
    __syncthreads();
    if (total == 0 && !A.bg && !(PRIM == 1 && count > 0)) {          // nothing can touch this tile: all outputs are zero
        if (wave != 0 || !inside) return;
        const int P0 = PS, pix0 = y * W + x;
        if (color) { float* o = color + (int64_t)b * 3 * P0 + pix0; o[0] = 0.f; o[P0] = 0.f; o[2 * P0] = 0.f; }
        if (mask) mask[(int64_t)b * P0 + pix0] = 0.f;
        if (depth) depth[(int64_t)b * P0 + pix0] = 0.f;
        if (normals) { float* o = normals + (int64_t)b * 3 * P0 + pix0; o[0] = 0.f; o[P0] = 0.f; o[2 * P0] = 0.f; }
        if (aux) {
            const unsigned gates0 = 127u;
            reinterpret_cast<float4*>(aux)[(int64_t)b * P0 + pix0] = make_float4(0.f, 0.f, 0.f, __uint_as_float(gates0));
        }
        return;
    }

    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (PRIM == 0) pixel_ray(A.Kinv + (int64_t)b * 9, (float)x, (float)y, rx, ry, rz);
    const float zn = (PRIM != 0) ? A.znorm[b] : 0.f;
    const float k00 = A.K[(int64_t)b * 9];
    float (*sdw)[64] = sd[wave];

    // ---- (2) walk the candidates in rounds of 64*SPL_NS, each cut into SPL_NS contiguous shares.  A wave stages its shares of a round in
    // its LDS slice (lane = candidate) -- all of them at once when they fit the slice, else share by share -- and broadcast-reads them.
    // A tile whose candidates all fit the slices stages once and keeps them for both sweeps.
    bool resident = false;
    auto stage = [&](int from, int n) {
        __syncthreads();
        if (lane < n) {
            const int s = overflow ? (from + lane) : lget(from + lane);
            const int64_t e = (sb + s) * 3;
            const float px = A.p_cam[e], py = A.p_cam[e + 1], pz = A.p_cam[e + 2];
            const float nx = A.n_cam[e], ny = A.n_cam[e + 1], nz = A.n_cam[e + 2];
            sdw[2][lane] = pz;
            sdw[3][lane] = nx; sdw[4][lane] = ny; sdw[5][lane] = nz;
            sdw[7][lane] = A.attr[e]; sdw[8][lane] = A.attr[e + 1]; sdw[9][lane] = A.attr[e + 2];
            if (PRIM == 0) {
                sdw[0][lane] = px; sdw[1][lane] = py;
                sdw[6][lane] = nx * px + ny * py + nz * pz;                  // :202
            } else {
                sdw[0][lane] = A.uv[(sb + s) * 2]; sdw[1][lane] = A.uv[(sb + s) * 2 + 1];
                sdw[6][lane] = fabsf(k00 * diam / (pz + FLT_EPSILON));       // :47 / :115
                sdw[10][lane] = depth_logit(pz, zn, C, nullptr);
            }
        }
        __syncthreads();
    };
    auto for_each = [&](auto&& body, auto&& quad) {
        for (int r0 = 0; r0 < total; r0 += 64 * SPL_NS) {
            const int nr = min(64 * SPL_NS, total - r0);
            const int q = (nr + SPL_NS - 1) / SPL_NS;                // share size (<= 64)
            const int base = r0 + wave * SPW * q;
            const int len = max(0, min(SPW * q, r0 + nr - base));    // this wave's candidates of the round
            const bool flat = (SPW == 1) || len <= 64;
            if (flat && !resident) stage(base, len);
#pragma unroll
            for (int j = 0; j < SPW; ++j) {
                const int c0w = base + j * q;
                const int kn = max(0, min(q, r0 + nr - c0w));
                if (!flat) stage(c0w, kn);
                const int ko = flat ? j * q : 0;
                int k = 0;
                for (; k + 3 < kn; k += 4) quad(j, ko + k, c0w + k);
                for (; k < kn; ++k) body(j, ko + k, c0w + k);
            }
        }
        resident = (SPW == 1) ? total <= 64 * SPL_NS : total <= 64;
    };

    float nu = 0.f, nue = 1.f;
    if (PRIM == 0) {   // per-pixel norm nu = || -t * mask ||_2 over the surfels  (:227-228)
        float part[SPW];
#pragma unroll
        for (int j = 0; j < SPW; ++j) part[j] = 0.f;
        auto nu_one = [&](int j, int k, int c) {
            const Hit h = disc_eval<ALT>(sdw[0][k], sdw[1][k], sdw[2][k], sdw[3][k], sdw[4][k], sdw[5][k], sdw[6][k], rx, ry, rz, A.cover_sq, diam, A.clamp_c);
            if (h.m) part[j] += h.t * h.t;
            if (use_cov) {
                const unsigned long long cm = __ballot(h.m);
                if (lane == 0) cov[c] = cm;
            }
        };
        // four candidates per step: an evaluation is ONE dependent chain of ~60 instructions (exact division included), and a dense tile -- the
        // reference's rendering_area 32: hundreds of candidates per share -- is paced by that latency with two waves per SIMD; four independent
        // chains in flight, the sums still taken in candidate order (same bits).  (The compiler does not unroll a loop around a ballot itself.)
        auto nu_quad = [&](int j, int k, int c) {
            Hit h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                h[u] = disc_eval<ALT>(sdw[0][k + u], sdw[1][k + u], sdw[2][k + u], sdw[3][k + u], sdw[4][k + u], sdw[5][k + u], sdw[6][k + u], rx, ry, rz,
                                      A.cover_sq, diam, A.clamp_c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (h[u].m) part[j] += h[u].t * h[u].t;
            if (use_cov) {
                unsigned long long cm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cm[u] = __ballot(h[u].m);
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) cov[c + u] = cm[u];
                }
            }
        };
        for_each(nu_one, nu_quad);
        float nu2;
        if (PW > 1) {
            nured[wave][lane] = part[0];
            __syncthreads();
            nu2 = nured[0][lane];
#pragma unroll
            for (int w = 1; w < PW; ++w) nu2 += nured[w][lane];
        } else {
            nu2 = part[0];
#pragma unroll
            for (int j = 1; j < SPW; ++j) nu2 += part[j];
        }
        nu = sqrtf(nu2);
        nue = nu + FLT_EPSILON;
    }
    // one sweep with a running maximum per share (online softmax): max logit, softmax sums and composites (rasterer.py:119-144)
    SplatAcc acc[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        acc[j].lmax = -FLT_MAX;
        acc[j].cs = 0.f; acc[j].c0 = 0.f; acc[j].c1 = 0.f; acc[j].c2 = 0.f; acc[j].dz = 0.f; acc[j].n0 = 0.f; acc[j].n1 = 0.f; acc[j].n2 = 0.f;
    }
    int ncov = 0;
    auto comp_one = [&](int j, int k, int c) {
        bool hit;
        float l;
        if (PRIM == 0) {
            if (use_cov) {
                const unsigned long long cm = cov[c];            // wave-uniform
                if (cm == 0ull) return;                           // covers no pixel of this tile
                hit = (cm >> lane) & 1ull;
                // the plane hit alone, with disc_eval's arithmetic (:209-211)
                const float b0 = rx * sdw[3][k] + ry * sdw[4][k] + rz * sdw[5][k];
                const float bb = (fabsf(b0) < 0.01f) ? FLT_EPSILON : b0;
                const float t = sdw[6][k] / bb;
                l = fmaxf((-t) / nue + 1.f, 0.f) * C;                             // :227-230
            } else {
                const Hit h = disc_eval<ALT>(sdw[0][k], sdw[1][k], sdw[2][k], sdw[3][k], sdw[4][k], sdw[5][k], sdw[6][k], rx, ry, rz, A.cover_sq, diam, A.clamp_c);
                l = fmaxf((-h.t) / nue + 1.f, 0.f) * C;
                hit = h.m;
            }
        } else if (PRIM == 1) {
            l = sdw[10][k];
            hit = circle_cover<ALT>(sdw[0][k], sdw[1][k], sdw[6][k], (float)x, (float)y, A.clamp_c);
        } else {
            l = sdw[10][k];
            hit = stamp_cover<ALT>(sdw[0][k], sdw[1][k], x, y, W, H, sdw[6][k]);
        }
        if (hit) {
            SplatAcc& a = acc[j];
            ++ncov;
            if (l > a.lmax) acc_rescale(a, l);
            const float e = expf(l - a.lmax);
            a.cs += e;
            a.c0 += e * sdw[7][k]; a.c1 += e * sdw[8][k]; a.c2 += e * sdw[9][k];
            a.dz += e * sdw[2][k];
            a.n0 += e * ((sdw[3][k] + 1.f) / 2.f); a.n1 += e * ((sdw[4][k] + 1.f) / 2.f); a.n2 += e * ((sdw[5][k] + 1.f) / 2.f);
        }
    };
    for_each(comp_one, [&](int j, int k, int c) { comp_one(j, k, c); comp_one(j, k + 1, c + 1); comp_one(j, k + 2, c + 2); comp_one(j, k + 3, c + 3); });
    // merge the shares' partial states, in share order, into share 0
    SplatAcc& S = acc[0];
    if (PW > 1) {
        float (*rw)[64] = red[wave];
        rw[0][lane] = S.lmax; rw[1][lane] = S.cs; rw[2][lane] = S.c0; rw[3][lane] = S.c1; rw[4][lane] = S.c2; rw[5][lane] = S.dz;
        rw[6][lane] = S.n0; rw[7][lane] = S.n1; rw[8][lane] = S.n2; rw[9][lane] = __int_as_float(ncov);
        __syncthreads();
        if (wave != 0) return;
        float M = S.lmax;
#pragma unroll
        for (int w = 1; w < PW; ++w) M = fmaxf(M, red[w][0][lane]);
        if (M > S.lmax) acc_rescale(S, M);
#pragma unroll
        for (int w = 1; w < PW; ++w) {
            const float f = expf(red[w][0][lane] - M);          // a share without hits holds lmax = -FLT_MAX and zero sums
            S.cs += f * red[w][1][lane]; S.c0 += f * red[w][2][lane]; S.c1 += f * red[w][3][lane]; S.c2 += f * red[w][4][lane];
            S.dz += f * red[w][5][lane]; S.n0 += f * red[w][6][lane]; S.n1 += f * red[w][7][lane]; S.n2 += f * red[w][8][lane];
            ncov += __float_as_int(red[w][9][lane]);
        }
    } else {
        float M = S.lmax;
#pragma unroll
        for (int j = 1; j < SPW; ++j) M = fmaxf(M, acc[j].lmax);
        if (M > S.lmax) acc_rescale(S, M);
#pragma unroll
        for (int j = 1; j < SPW; ++j) {
            const float f = expf(acc[j].lmax - M);
            S.cs += f * acc[j].cs; S.c0 += f * acc[j].c0; S.c1 += f * acc[j].c1; S.c2 += f * acc[j].c2;
            S.dz += f * acc[j].dz; S.n0 += f * acc[j].n0; S.n1 += f * acc[j].n1; S.n2 += f * acc[j].n2;
        }
    }
    float lmax = S.lmax, cs = S.cs, c0 = S.c0, c1 = S.c1, c2 = S.c2, dz = S.dz, n0 = S.n0, n1 = S.n1, n2 = S.n2;
    auto rescale = [&](float newmax) {
        const float f = expf(lmax - newmax);
        cs *= f; c0 *= f; c1 *= f; c2 *= f; dz *= f; n0 *= f; n1 *= f; n2 *= f;
        lmax = newmax;
    };
    const int nunc = count - ncov;
    if (PRIM == 1 && nunc > 0 && 0.f > lmax) rescale(0.f);                        // uncovered surfels keep logit 0 (:70)
    const float lbg = A.bg ? A.bg_logit[b] : 0.f;
    if (A.bg && lbg > lmax) rescale(lbg);
    if (!inside) return;
    const int P = PS;
    const int pix = y * W + x;
    float den = cs;
    if (PRIM == 1 && nunc > 0) den += (float)nunc * expf(0.f - lmax);
    if (A.bg) {
        const float eb = expf(lbg - lmax);
        den += eb;
        cs += eb;
        const float* bgp = A.bg + (int64_t)b * 3 * P + pix;
        c0 += eb * bgp[0]; c1 += eb * bgp[P]; c2 += eb * bgp[2 * P];
    }
    const bool act = den > 0.f;
    const float inv = act ? 1.f / den : 0.f;
    c0 *= inv; c1 *= inv; c2 *= inv; dz *= inv; n0 *= inv; n1 *= inv; n2 *= inv;
    const float ms = (act && cs == den) ? 1.f : cs * inv;     // every row covered or background: the weights sum to one
    unsigned gates = 0;
    gates |= (c0 <= 1.f) ? 1u : 0u; gates |= (c1 <= 1.f) ? 2u : 0u; gates |= (c2 <= 1.f) ? 4u : 0u;
    gates |= (ms <= 1.f) ? 8u : 0u;
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
        a4.x = nu; a4.y = act ? lmax : 0.f; a4.z = den; a4.w = __uint_as_float(gates);
        reinterpret_cast<float4*>(aux)[(int64_t)b * P + pix] = a4;
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------

// Sum over the 64 lanes of a wave, the same value in every lane.  r06: DPP row shifts + row broadcasts (one v_add_f32_dpp per step, no LDS crossbar):
// lane 63 ends up with  ((row0) + (row1)) + ((row2) + (row3)),  each row summed as a shifted prefix -- a fixed order, bit-repeatable, but not
// r05's butterfly order (__shfl_xor: 6 ds_bpermute + 6 adds per value, 12 values per surfel = a sixth of the backward's instructions).
// 1 / x to within an ulp: hardware reciprocal estimate + one Newton step (for gradient arithmetic only, see the chain below)
__device__ __forceinline__ float fast_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);        // row_shr:1
    v = dpp_add<0x112, 0xf>(v);        // row_shr:2
    v = dpp_add<0x114, 0xf>(v);        // row_shr:4
    v = dpp_add<0x118, 0xf>(v);        // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// DENSE: the upstream gradient is given per (surfel, pixel) weight -- gW [B][rows][P], rows = cap (+1 with a background row) -- together with
// Sd[b][pix] = sum_j w_j gW_j (the softmax-backward sum), instead of through the composited images: the backward of the standalone
// primitives (sdfr_splat_weights), which hand out the dense weight matrix as the reference's inside_* functions do.
// KS (r06): the upstream colour gradient arrives UN-normalised with a per-crop factor, kscale[2 b] (sdfr_losses_fused): g_color * k on load --
// the product the 2-D loss's finalize pass used to store in a sweep of its own.
template <int PRIM, bool DENSE = false, bool ALT = false, bool KS = false>
__global__ __launch_bounds__(256) void sdfr_splat_bwd_kernel(const SplatArgs A, const float* __restrict__ aux,
                                                            const float* __restrict__ color, const float* __restrict__ mask,
                                                            const float* __restrict__ depth, const float* __restrict__ normals,
                                                            const float* __restrict__ g_color, const float* __restrict__ g_mask,
                                                            const float* __restrict__ g_depth, const float* __restrict__ g_normals,
                                                            float* __restrict__ g_p, float* __restrict__ g_n, float* __restrict__ g_attr,
                                                            const float* __restrict__ gW = nullptr, const float* __restrict__ Sd = nullptr,
                                                            int rows = 0, const float* __restrict__ kscale = nullptr,
                                                            const int4* __restrict__ boxes = nullptr) {
    int xb, b;
    sdfr_xcd_crop_map(xb, b);          // a crop's surfels on one XCD: its pixel records (aux, images, upstream gradients) are fetched by one L2
    const float kc = KS ? kscale[2 * b] : 1.f;
    const int lane = threadIdx.x & 63;
    const int s = xb * 4 + (threadIdx.x >> 6);
    if (s >= sdfr_count(A.cnt, b, A.cap)) return;
    int W, H, PS;
    splat_dims(A, b, W, H, PS);
    const float diam = A.diam, C = A.depth_constant;
    const int64_t e1 = (int64_t)b * A.cap + s;
    const int64_t e = e1 * 3;
    const float px = A.p_cam[e], py = A.p_cam[e + 1], pz = A.p_cam[e + 2];
    const float nx = A.n_cam[e], ny = A.n_cam[e + 1], nz = A.n_cam[e + 2];
    const float a0 = A.attr[e], a1 = A.attr[e + 1], a2 = A.attr[e + 2];
    const float m0 = (nx + 1.f) / 2.f, m1 = (ny + 1.f) / 2.f, m2 = (nz + 1.f) / 2.f;
    const float a = nx * px + ny * py + nz * pz;
    const float* Ki = A.Kinv + (int64_t)b * 9;
    const int P = PS;
    // PRIM 1,2: pixel-independent logit
    float u = 0.f, v = 0.f, rad = 0.f, zl = 0.f, q0 = 0.f, zn = 0.f;
    if (PRIM != 0) {
        u = A.uv[e1 * 2]; v = A.uv[e1 * 2 + 1];
        zn = A.znorm[b];
        rad = fabsf(A.K[(int64_t)b * 9] * diam / (pz + FLT_EPSILON));
        zl = depth_logit(pz, zn, C, &q0);
    }
    int x0, y0, x1, y1;
    float sC0 = 0.f, sC1 = 0.f, sC2 = 0.f, sN0 = 0.f, sN1 = 0.f, sN2 = 0.f, sZ = 0.f, sA = 0.f, sB0 = 0.f, sB1 = 0.f, sB2 = 0.f, sL = 0.f;
    // r05: two phases per surfel.  The coverage test is cheap and most lanes of a box fail it (a disc of radius 4 px covers ~53 of the ~170
    // pixels of its conservative box), while the gradient chain behind it is expensive and ran for a whole wave iteration whenever ONE lane
    // was covered.  So: scan the box 64 pixels at a time, append the covered pixels (position, plane hit) to a per-wave LDS queue with a
    // ballot, and run the gradient chain on the queue -- full lanes, one to two iterations per surfel instead of three to six.  The queue is
    // drained whenever fewer than 64 slots are free, so any box size works.  Sums are taken in queue order: deterministic, bit-repeatable.
    __shared__ float4 queue[4][SPL_BQ];
    float4* qw = queue[threadIdx.x >> 6];
    auto chain = [&](int x, int y, float ht, float hb) {
        const int pix = y * W + x;
        const float4 ax = reinterpret_cast<const float4*>(aux)[(int64_t)b * P + pix];
        const float nue = ax.x + FLT_EPSILON;
        const unsigned gates = __float_as_uint(ax.w);
        float q = q0, logit = zl;
        if (PRIM == 0) {
            q = (-ht) / nue + 1.f;
            logit = fmaxf(q, 0.f) * C;
        }
        // (r06: the divisions of the GRADIENT chain -- softmax normalisation, d zeta / d t, d t / d a, d t / d b -- take v_rcp_f32 + one Newton step,
        // within 1 ulp of the IEEE quotient; their results are compared against 1e-3 relative, never against a threshold.  Every division that
        // feeds a DECISION -- the plane hit t behind the coverage test, q behind the clamp gate -- stays exact.)
        const float w = expf(logit - ax.y) * fast_rcp(ax.z);
        // gated upstream gradients and S = sum_j w_j dL/dw_j = <gated grads, composited outputs>
        float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, gm = 0.f, gd = 0.f, gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, S = 0.f;
        if (DENSE) S = Sd[(int64_t)b * P + pix];
        if (!DENSE && g_color) {
            const float* g = g_color + (int64_t)b * 3 * P + pix;
            const float* o = color + (int64_t)b * 3 * P + pix;
            if (KS) { gc0 = (gates & 1u) ? g[0] * kc : 0.f; gc1 = (gates & 2u) ? g[P] * kc : 0.f; gc2 = (gates & 4u) ? g[2 * P] * kc : 0.f; }
            else { gc0 = (gates & 1u) ? g[0] : 0.f; gc1 = (gates & 2u) ? g[P] : 0.f; gc2 = (gates & 4u) ? g[2 * P] : 0.f; }
            S += gc0 * o[0] + gc1 * o[P] + gc2 * o[2 * P];
        }
        if (!DENSE && g_mask) { gm = (gates & 8u) ? g_mask[(int64_t)b * P + pix] : 0.f; S += gm * mask[(int64_t)b * P + pix]; }
        if (!DENSE && g_depth) { gd = g_depth[(int64_t)b * P + pix]; S += gd * depth[(int64_t)b * P + pix]; }
        if (!DENSE && g_normals) {
            const float* g = g_normals + (int64_t)b * 3 * P + pix;
            const float* o = normals + (int64_t)b * 3 * P + pix;
            gn0 = (gates & 16u) ? g[0] : 0.f; gn1 = (gates & 32u) ? g[P] : 0.f; gn2 = (gates & 64u) ? g[2 * P] : 0.f;
            S += gn0 * o[0] + gn1 * o[P] + gn2 * o[2 * P];
        }
        const float dLdw = DENSE ? gW[((int64_t)b * rows + s) * P + pix]
                                 : gc0 * a0 + gc1 * a1 + gc2 * a2 + gm + gd * pz + gn0 * m0 + gn1 * m1 + gn2 * m2;
        sC0 += w * gc0; sC1 += w * gc1; sC2 += w * gc2;
        sN0 += w * gn0; sN1 += w * gn1; sN2 += w * gn2;
        sZ += w * gd;
        const float dl = w * (dLdw - S);
        if (PRIM == 0) {
            const float dq = (q >= 0.f) ? dl * C : 0.f;
            const float dt = -(dq * fast_rcp(nue));                 // zeta = -t * mask
            const float ihb = fast_rcp(hb);
            sA += dt * ihb;                                         // t = a / b
            if (hb != FLT_EPSILON) {                                // (|n.ray| < 0.01 was replaced by eps, :210: no gradient through b)
                float rx, ry, rz;
                pixel_ray(Ki, (float)x, (float)y, rx, ry, rz);
                const float db = -dt * ht * ihb;
                sB0 += db * rx; sB1 += db * ry; sB2 += db * rz;
            }
        } else {
            sL += dl;
        }
    };
    auto drain = [&](int count) {
        for (int e = lane; e < count; e += 64) {
            const float4 en = qw[e];
            const unsigned xy = __float_as_uint(en.x);
            chain((int)(xy & 0xffffu), (int)(xy >> 16), en.y, en.z);
        }
    };
    // (r06: the forward pass of the same step has written every surfel's conservative screen box into the splat workspace -- the same disc_bbox --;
    // a caller that hands it over saves the ~80 instructions per surfel of recomputing it: two square roots, four divisions)
    bool onscreen;
    if (boxes) {
        const int4 bb = boxes[e1];
        x0 = max(bb.x, 0); y0 = max(bb.y, 0); x1 = min(bb.z, W - 1); y1 = min(bb.w, H - 1);      // (never beyond this crop's extent, whatever the workspace holds)
        onscreen = x0 <= x1 && y0 <= y1;
    } else onscreen = surfel_bbox<PRIM, ALT>(A, b, e1, x0, y0, x1, y1);
    if (onscreen) {
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1, npx = bw * bh;
        int nq = 0;                                                  // wave-uniform fill of the queue
        for (int base = 0; base < npx; base += 64) {
            const int i = base + lane;
            bool cov = false;
            int x = 0, y = 0;
            float ht = 0.f, hb = 1.f;
            if (i < npx) {
                const int yy = i / bw;
                x = x0 + (i - yy * bw); y = y0 + yy;
                if (PRIM == 0) {
                    float rx, ry, rz;
                    pixel_ray(Ki, (float)x, (float)y, rx, ry, rz);
                    const Hit h = disc_eval<ALT>(px, py, pz, nx, ny, nz, a, rx, ry, rz, A.cover_sq, diam, A.clamp_c);
                    cov = h.m; ht = h.t; hb = h.b;
                } else if (PRIM == 1) {
                    cov = circle_cover<ALT>(u, v, rad, (float)x, (float)y, A.clamp_c);
                } else {
                    cov = stamp_cover<ALT>(u, v, x, y, W, H, rad);
                }
            }
            const unsigned long long bal = __ballot(cov);
            if (cov) qw[nq + __popcll(bal & ((1ull << lane) - 1ull))] = make_float4(__uint_as_float((unsigned)x | ((unsigned)y << 16)), ht, hb, 0.f);
            nq += __popcll(bal);
            // (the queue is written and read by ONE wave, other lanes' entries: a wave barrier keeps the compiler from moving the reads above
            // the writes; the hardware executes a wave's LDS operations in order)
            if (nq > SPL_BQ - 64) { __builtin_amdgcn_wave_barrier(); drain(nq); nq = 0; __builtin_amdgcn_wave_barrier(); }
        }
        __builtin_amdgcn_wave_barrier();
        drain(nq);
    }
    sC0 = wave_sum(sC0); sC1 = wave_sum(sC1); sC2 = wave_sum(sC2);
    sN0 = wave_sum(sN0); sN1 = wave_sum(sN1); sN2 = wave_sum(sN2);
    sZ = wave_sum(sZ);
    if (PRIM == 0) { sA = wave_sum(sA); sB0 = wave_sum(sB0); sB1 = wave_sum(sB1); sB2 = wave_sum(sB2); }
    else sL = wave_sum(sL);
    if (lane == 0) {
        if (g_attr) { g_attr[e] = sC0; g_attr[e + 1] = sC1; g_attr[e + 2] = sC2; }
        if (PRIM == 0) {
            g_n[e] = 0.5f * sN0 + sB0 + sA * px;
            g_n[e + 1] = 0.5f * sN1 + sB1 + sA * py;
            g_n[e + 2] = 0.5f * sN2 + sB2 + sA * pz;
            g_p[e] = sA * nx;
            g_p[e + 1] = sA * ny;
            g_p[e + 2] = sA * nz + sZ;
        } else {
            g_n[e] = 0.5f * sN0; g_n[e + 1] = 0.5f * sN1; g_n[e + 2] = 0.5f * sN2;
            const float dq = (q0 >= 0.f) ? sL * C : 0.f;              // logit = clamp(q,0)*C, q = -z/(zn+eps) + 1, zn detached
            g_p[e] = 0.f; g_p[e + 1] = 0.f;
            g_p[e + 2] = sZ - dq / (zn + FLT_EPSILON);
        }
    }
}

// ---- dense weight matrix (the standalone primitives) ---------------------------------------------------------------------------
// The reference's inside_surfel / inside_circle / inside_circle_opt RETURN the (N[+1], P) weight matrix (primitives.py:71,162,243).
// Rasterer.forward never needs it here (the splat kernels composite on the fly), but a caller of the standalone functions does: the
// per-pixel softmax state `aux` of a splat forward pass plus one more sweep over each surfel's screen box reproduces every entry.
// One wavefront per surfel; W must be zero-filled by the caller (only covered pixels are written).
template <int PRIM, bool ALT = false>
__global__ __launch_bounds__(256) void sdfr_splat_weights_kernel(const SplatArgs A, const float* __restrict__ aux, float* __restrict__ Wout,
                                                                int rows) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= sdfr_count(A.cnt, b, A.cap)) return;
    const int W = A.W, H = A.H, P = W * H;
    const int64_t e1 = (int64_t)b * A.cap + s, e = e1 * 3;
    const float px = A.p_cam[e], py = A.p_cam[e + 1], pz = A.p_cam[e + 2];
    const float nx = A.n_cam[e], ny = A.n_cam[e + 1], nz = A.n_cam[e + 2];
    const float a = nx * px + ny * py + nz * pz;
    const float* Ki = A.Kinv + (int64_t)b * 9;
    float u = 0.f, v = 0.f, rad = 0.f, zl = 0.f;
    if (PRIM != 0) {
        u = A.uv[e1 * 2]; v = A.uv[e1 * 2 + 1];
        rad = fabsf(A.K[(int64_t)b * 9] * A.diam / (pz + FLT_EPSILON));
        zl = depth_logit(pz, A.znorm[b], A.depth_constant, nullptr);
    }
    int x0, y0, x1, y1;
    if (!surfel_bbox<PRIM, ALT>(A, b, e1, x0, y0, x1, y1)) return;
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    float* row = Wout + ((int64_t)b * rows + s) * P;
    for (int i = lane; i < bw * bh; i += 64) {
        const int yy = i / bw;
        const int x = x0 + (i - yy * bw), y = y0 + yy;
        bool cov;
        float logit = zl;
        const int pix = y * W + x;
        const float4 ax = reinterpret_cast<const float4*>(aux)[(int64_t)b * P + pix];
        if (PRIM == 0) {
            float rx, ry, rz;
            pixel_ray(Ki, (float)x, (float)y, rx, ry, rz);
            const Hit h = disc_eval<ALT>(px, py, pz, nx, ny, nz, a, rx, ry, rz, A.cover_sq, A.diam, A.clamp_c);
            cov = h.m;
            logit = fmaxf((-h.t) / (ax.x + FLT_EPSILON) + 1.f, 0.f) * A.depth_constant;
        } else if (PRIM == 1) {
            cov = circle_cover<ALT>(u, v, rad, (float)x, (float)y, A.clamp_c);
        } else {
            cov = stamp_cover<ALT>(u, v, x, y, W, H, rad);
        }
        if (cov) row[pix] = expf(logit - ax.y) / ax.z;
    }
}

__global__ __launch_bounds__(256) void sdfr_splat_bgweights_kernel(const float* __restrict__ aux, const float* __restrict__ bg_logit,
                                                                  float* __restrict__ Wout, int rows, int P) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= P) return;
    const float4 ax = reinterpret_cast<const float4*>(aux)[(int64_t)b * P + pix];
    Wout[((int64_t)b * rows + (rows - 1)) * P + pix] = expf(bg_logit[b] - ax.y) / ax.z;
}

// ---- C ABI --------------------------------------------------------------------------------------------------------

static int fill_args(SplatArgs& A, const char* who, int primitive, const float* K, const float* Kinv, const float* p_cam,
                     const float* n_cam, const float* attr, const float* uv, const float* znorm, const float* bg, const float* bg_logit,
                     int B, int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant) {
    SDFR_REQUIRE(primitive >= 0 && primitive <= 2, "%s: primitive %d unknown (0 disc, 1 circle, 2 circle_opt)", who, primitive);
    SDFR_REQUIRE(K && Kinv, "%s: NULL intrinsics", who);
    SDFR_REQUIRE(W > 0 && H > 0 && B >= 0 && cap >= 0, "%s: bad size", who);
    SDFR_REQUIRE(W < 65536 && H < 65536, "%s: images of up to 65535 x 65535 pixels (the backward queues pixel coordinates as 16-bit pairs)", who);
    SDFR_REQUIRE(cap == 0 || (p_cam && n_cam && attr), "%s: NULL surfel array", who);
    SDFR_REQUIRE(primitive == 0 || cap == 0 || (uv && znorm), "%s: circle primitives need uv and znorm", who);
    SDFR_REQUIRE((bg == nullptr) == (bg_logit == nullptr), "%s: bg and bg_logit go together", who);
    A.K = K; A.Kinv = Kinv; A.p_cam = p_cam; A.n_cam = n_cam; A.attr = attr; A.uv = uv; A.znorm = znorm; A.bg = bg; A.bg_logit = bg_logit;
    A.cap = cap; A.cnt = cnt; A.W = W; A.H = H; A.diam = diam; A.depth_constant = depth_constant;
    A.cover_sq = disc_cover_sq(diam);
    A.clamp_c = (primitive == 1) ? 3.f : 5.f;                 // the functions' default softclamp_constant (primitives.py:13,83,175)
    A.reach = SIGMOID_REACH;
    A.wh = nullptr; A.pst = W * H; A.bin_stride = sdfr_splat_bin_stride(cap, W, H);
    return SDFR_OK;
}

// the clamp configuration of the *_clamp entry points: clamp_alt = the primitive's OTHER clamp (disc: softclamp=True; circle, circle_opt:
// softclamp=False), clamp_constant = softclamp_constant of the sigmoid clamps (> 0)
static int set_clamp(SplatArgs& A, const char* who, int primitive, int clamp_alt, float clamp_constant) {
    SDFR_REQUIRE(clamp_alt == 0 || clamp_alt == 1, "%s: clamp_alt %d (0: the renderer's clamp of the primitive, 1: the other one)", who, clamp_alt);
    const bool sigmoid = (primitive == 0) ? clamp_alt == 1 : clamp_alt == 0;
    if (sigmoid) {
        SDFR_REQUIRE(clamp_constant > 0.f && clamp_constant < 1e30f, "%s: softclamp_constant %g must be positive", who, (double)clamp_constant);
        A.clamp_c = clamp_constant;
        // sigmoid(x) > 0 in float32 while exp(-x) is finite: x > -88.73; beyond the radius by less than 88.73 / c (+ slack for the rounding of
        // the product).  c = 3 keeps the constant the renderer path was built and tested with.
        A.reach = (clamp_constant == 3.f) ? SIGMOID_REACH : 88.73f / clamp_constant * 1.001f + 0.05f;
    }
    return SDFR_OK;
}

// tiles per launch from which the forward runs one wave per tile (SDFR_SPLAT_SERIAL_TILES overrides: 0 = always, a huge value = never).
// Measured at 256x256 (1024 tiles per crop), lists ready: 4 crops 44 us wide / 129 us serial (the heaviest tile's chain is the launch),
// 16 crops 146 / 157, 64 crops 550 / 424; with the square-root-free coverage test: 16 crops 136 / 132, 32 crops 261 / 211, 64 crops - / 360.
static int64_t splat_serial_tiles() {
    static const int64_t v = [] {
        const char* e = getenv("SDFR_SPLAT_SERIAL_TILES");
        return e ? (int64_t)atoll(e) : (int64_t)16384;
    }();
    return v;
}

static int splat_forward_launch(const SplatArgs& A, int primitive, bool boxes_ready, bool use_bins, int B, int cap, int tiles, int32_t* bbox_ws,
                                float* color, float* mask, float* depth, float* normals, float* aux, void* stream, bool alt = false) {
    SDFR_REQUIRE(cap == 0 || bbox_ws, "sdfr_splat_forward: NULL bbox workspace");
    if (B == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    int4* bb = reinterpret_cast<int4*>(bbox_ws);
    // SDFR_PRIM_BINS: tile lists behind the boxes (splat_bbox.h); otherwise the workspace holds the boxes only and every tile scans all boxes
    int32_t* bins = (cap > 0 && use_bins) ? bbox_ws + (int64_t)B * cap * 4 : nullptr;
    const dim3 gb(sdfr_cdiv(cap > 0 ? cap : 1, 256), B);
    const dim3 gt(tiles, B);
    // launch geometry (same results bit for bit): one wave per candidate share while the tiles do not fill the chip, one wave per tile beyond
    const bool serial = (int64_t)gt.x * B >= splat_serial_tiles();
#define SPL_LAUNCH_FWD(P)                                                                                                              \
    do {                                                                                                                               \
        if (cap > 0 && !boxes_ready) hipLaunchKernelGGL(sdfr_splat_bbox_kernel<P>, gb, dim3(256), 0, s, A, bb);                        \
        if (bins && !boxes_ready) hipLaunchKernelGGL(sdfr_splat_bin_kernel, dim3(B), dim3(1024), 0, s, A, bb, bins);                   \
        if (serial) hipLaunchKernelGGL((sdfr_splat_fwd_kernel<P, 1>), gt, dim3(64), 0, s, A, bb, bins, color, mask, depth, normals, aux); \
        else hipLaunchKernelGGL((sdfr_splat_fwd_kernel<P, SPL_NS>), gt, dim3(64 * SPL_NS), 0, s, A, bb, bins, color, mask, depth, normals, aux); \
    } while (0)
    // the other clamp of the primitive (standalone functions only): one wave per tile, boxes computed here (never boxes_ready / binned)
#define SPL_LAUNCH_FWD_ALT(P)                                                                                                          \
    do {                                                                                                                               \
        if (cap > 0) hipLaunchKernelGGL((sdfr_splat_bbox_kernel<P, true>), gb, dim3(256), 0, s, A, bb);                                \
        hipLaunchKernelGGL((sdfr_splat_fwd_kernel<P, 1, true>), gt, dim3(64), 0, s, A, bb, (const int32_t*)nullptr, color, mask, depth, normals, aux); \
    } while (0)
    if (alt) {
        SDFR_REQUIRE(!boxes_ready && !use_bins, "sdfr_splat_forward_clamp: the alternative clamp computes its own screen boxes");
        switch (primitive) {
            case 0: SPL_LAUNCH_FWD_ALT(0); break;
            case 1: SPL_LAUNCH_FWD_ALT(1); break;
            default: SPL_LAUNCH_FWD_ALT(2); break;
        }
    } else
    switch (primitive) {
        case 0: SPL_LAUNCH_FWD(0); break;
        case 1: SPL_LAUNCH_FWD(1); break;
        default: SPL_LAUNCH_FWD(2); break;
    }
#undef SPL_LAUNCH_FWD
#undef SPL_LAUNCH_FWD_ALT
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_splat_forward(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam,
                                  const float* attr, const float* uv, const float* znorm, const float* bg, const float* bg_logit, int B,
                                  int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant, int32_t* bbox_ws,
                                  float* color, float* mask, float* depth, float* normals, float* aux, void* stream) {
    const int prim = primitive & ~(SDFR_PRIM_BOXES_READY | SDFR_PRIM_BINS);
    return sdfr_splat_forward_clamp(primitive, K, Kinv, p_cam, n_cam, attr, uv, znorm, bg, bg_logit, B, cap, cnt, W, H, diam, depth_constant, 0,
                                    prim == 1 ? 3.f : 5.f, bbox_ws, color, mask, depth, normals, aux, stream);
}

// ... with the primitive's clamp configuration spelled out (r05): what the standalone inside_* functions accept beyond the renderer's calls
extern "C" int sdfr_splat_forward_clamp(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam,
                                        const float* attr, const float* uv, const float* znorm, const float* bg, const float* bg_logit, int B,
                                        int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant, int clamp_alt,
                                        float clamp_constant, int32_t* bbox_ws, float* color, float* mask, float* depth, float* normals,
                                        float* aux, void* stream) {
    const bool boxes_ready = (primitive & SDFR_PRIM_BOXES_READY) != 0;      // bbox_ws already holds boxes and tile lists (sdfr_surfels_forward)
    const bool use_bins = (primitive & SDFR_PRIM_BINS) != 0;
    primitive &= ~(SDFR_PRIM_BOXES_READY | SDFR_PRIM_BINS);
    SplatArgs A;
    int rc = fill_args(A, "sdfr_splat_forward", primitive, K, Kinv, p_cam, n_cam, attr, uv, znorm, bg, bg_logit, B, cap, cnt, W, H, diam,
                       depth_constant);
    if (rc) return rc;
    rc = set_clamp(A, "sdfr_splat_forward_clamp", primitive, clamp_alt, clamp_constant);
    if (rc) return rc;
    return splat_forward_launch(A, primitive, boxes_ready, use_bins, B, cap, ((W + 7) / 8) * ((H + 7) / 8), bbox_ws, color, mask, depth, normals,
                                aux, stream, clamp_alt != 0);
}

// ---- ragged extents: every crop of the batch its own image size (W_b, H_b) and intrinsics, read from device memory ------------------------
// wh int32[B][2] (device); images / aux in slots of pix_stride pixels per channel; tiles_cap >= ceil(W_b/8) ceil(H_b/8) for every crop
// (launch bound and tile-list layout).  One captured launch sequence serves any crop sizes within the caps.
extern "C" int64_t sdfr_splat_ws_words_r(int B, int cap, int tiles_cap) {
    return (int64_t)B * ((int64_t)cap * 4 + (int64_t)tiles_cap + 2 + (int64_t)SPL_LM * cap);
}

static int fill_args_r(SplatArgs& A, const char* who, const float* K, const float* Kinv, const float* p_cam, const float* n_cam,
                       const float* attr, int B, int cap, const int32_t* cnt, const int32_t* wh, int pix_stride, int tiles_cap, float diam,
                       float depth_constant) {
    int rc = fill_args(A, who, 0, K, Kinv, p_cam, n_cam, attr, nullptr, nullptr, nullptr, nullptr, B, cap, cnt, 1, 1, diam, depth_constant);
    if (rc) return rc;
    SDFR_REQUIRE(wh && pix_stride > 0 && tiles_cap > 0, "%s: ragged extents need wh, pix_stride and tiles_cap", who);
    A.wh = wh; A.pst = pix_stride; A.W = 0; A.H = 0;
    A.bin_stride = (int64_t)tiles_cap + 2 + (int64_t)SPL_LM * cap;
    return SDFR_OK;
}

extern "C" int sdfr_splat_forward_r(int flags, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr,
                                    int B, int cap, const int32_t* cnt, const int32_t* wh, int pix_stride, int tiles_cap, float diam,
                                    float depth_constant, int32_t* bbox_ws, float* color, float* mask, float* depth, float* normals, float* aux,
                                    void* stream) {
    const bool boxes_ready = (flags & SDFR_PRIM_BOXES_READY) != 0;
    const bool use_bins = (flags & SDFR_PRIM_BINS) != 0;
    SDFR_REQUIRE((flags & ~(SDFR_PRIM_BOXES_READY | SDFR_PRIM_BINS)) == 0, "sdfr_splat_forward_r: disc primitive only (flags = BOXES_READY | BINS)");
    SplatArgs A;
    int rc = fill_args_r(A, "sdfr_splat_forward_r", K, Kinv, p_cam, n_cam, attr, B, cap, cnt, wh, pix_stride, tiles_cap, diam, depth_constant);
    if (rc) return rc;
    return splat_forward_launch(A, 0, boxes_ready, use_bins, B, cap, tiles_cap, bbox_ws, color, mask, depth, normals, aux, stream);
}

extern "C" int sdfr_splat_backward_r(const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr, int B, int cap,
                                     const int32_t* cnt, const int32_t* wh, int pix_stride, float diam, float depth_constant, const float* aux,
                                     const float* color, const float* mask, const float* depth, const float* normals, const float* g_color,
                                     const float* g_mask, const float* g_depth, const float* g_normals, float* g_p_cam, float* g_n_cam,
                                     float* g_attr, void* stream) {
    SplatArgs A;
    int rc = fill_args_r(A, "sdfr_splat_backward_r", K, Kinv, p_cam, n_cam, attr, B, cap, cnt, wh, pix_stride, 1, diam, depth_constant);
    if (rc) return rc;
    SDFR_REQUIRE(aux && g_p_cam && g_n_cam && g_attr, "sdfr_splat_backward_r: NULL argument");
    SDFR_REQUIRE((!g_color || color) && (!g_mask || mask) && (!g_depth || depth) && (!g_normals || normals),
                 "sdfr_splat_backward_r: an image gradient was given without the forward image");
    if (B == 0 || cap == 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_splat_bwd_kernel<0>, dim3(sdfr_cdiv(cap, 4), B), dim3(256), 0, (hipStream_t)stream, A, aux, color, mask, depth, normals,
                       g_color, g_mask, g_depth, g_normals, g_p_cam, g_n_cam, g_attr);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// The disc primitive's backward with a per-crop factor on the colour gradient (r06): g_color holds the un-normalised 2-D loss gradient and
// kscale float[B][2] the factors sdfr_losses_fused published (kscale[2 b] applies).  wh == NULL: dense W x H images; else ragged extents.
// bbox_ws (may be NULL): the splat workspace of this step's sdfr_surfels_forward(_r) / sdfr_splat_forward(_r) -- its head holds the surfels' screen
// boxes [B][cap][4], which are then read instead of recomputed.
extern "C" int sdfr_splat_backward_x(const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr, int B, int cap,
                                     const int32_t* cnt, int W, int H, const int32_t* wh, int pix_stride, float diam, float depth_constant,
                                     const float* aux, const float* color, const float* g_color, const float* kscale, float* g_p_cam,
                                     float* g_n_cam, float* g_attr, const int32_t* bbox_ws, void* stream) {
    SplatArgs A;
    int rc = wh ? fill_args_r(A, "sdfr_splat_backward_x", K, Kinv, p_cam, n_cam, attr, B, cap, cnt, wh, pix_stride, 1, diam, depth_constant)
                : fill_args(A, "sdfr_splat_backward_x", 0, K, Kinv, p_cam, n_cam, attr, nullptr, nullptr, nullptr, nullptr, B, cap, cnt, W, H, diam,
                            depth_constant);
    if (rc) return rc;
    SDFR_REQUIRE(aux && color && g_color && kscale && g_p_cam && g_n_cam && g_attr, "sdfr_splat_backward_x: NULL argument");
    if (B == 0 || cap == 0) return SDFR_OK;
    hipLaunchKernelGGL((sdfr_splat_bwd_kernel<0, false, false, true>), dim3(sdfr_cdiv(cap, 4), B), dim3(256), 0, (hipStream_t)stream, A, aux, color,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, g_color, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, g_p_cam, g_n_cam, g_attr, (const float*)nullptr, (const float*)nullptr, 0, kscale,
                       reinterpret_cast<const int4*>(bbox_ws));
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_splat_backward(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam,
                                   const float* attr, const float* uv, const float* znorm, const float* bg, const float* bg_logit, int B,
                                   int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant, const float* aux,
                                   const float* color, const float* mask, const float* depth, const float* normals,
                                   const float* g_color, const float* g_mask, const float* g_depth, const float* g_normals,
                                   float* g_p_cam, float* g_n_cam, float* g_attr, void* stream) {
    SplatArgs A;
    int rc = fill_args(A, "sdfr_splat_backward", primitive, K, Kinv, p_cam, n_cam, attr, uv, znorm, bg, bg_logit, B, cap, cnt, W, H, diam,
                       depth_constant);
    if (rc) return rc;
    SDFR_REQUIRE(aux && g_p_cam && g_n_cam && g_attr, "sdfr_splat_backward: NULL argument");
    SDFR_REQUIRE((!g_color || color) && (!g_mask || mask) && (!g_depth || depth) && (!g_normals || normals),
                 "sdfr_splat_backward: an image gradient was given without the forward image");
    if (B == 0 || cap == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(sdfr_cdiv(cap, 4), B);
    switch (primitive) {
        case 0: hipLaunchKernelGGL(sdfr_splat_bwd_kernel<0>, g, dim3(256), 0, s, A, aux, color, mask, depth, normals, g_color, g_mask,
                                   g_depth, g_normals, g_p_cam, g_n_cam, g_attr); break;
        case 1: hipLaunchKernelGGL(sdfr_splat_bwd_kernel<1>, g, dim3(256), 0, s, A, aux, color, mask, depth, normals, g_color, g_mask,
                                   g_depth, g_normals, g_p_cam, g_n_cam, g_attr); break;
        default: hipLaunchKernelGGL(sdfr_splat_bwd_kernel<2>, g, dim3(256), 0, s, A, aux, color, mask, depth, normals, g_color, g_mask,
                                    g_depth, g_normals, g_p_cam, g_n_cam, g_attr); break;
    }
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_splat_weights(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* uv,
                                  const float* znorm, const float* bg_logit, int B, int cap, const int32_t* cnt, int W, int H, float diam,
                                  float depth_constant, const float* aux, float* weights, void* stream) {
    return sdfr_splat_weights_clamp(primitive, K, Kinv, p_cam, n_cam, uv, znorm, bg_logit, B, cap, cnt, W, H, diam, depth_constant, 0,
                                    primitive == 1 ? 3.f : 5.f, aux, weights, stream);
}

extern "C" int sdfr_splat_weights_clamp(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* uv,
                                        const float* znorm, const float* bg_logit, int B, int cap, const int32_t* cnt, int W, int H, float diam,
                                        float depth_constant, int clamp_alt, float clamp_constant, const float* aux, float* weights,
                                        void* stream) {
    SplatArgs A;
    int rc = fill_args(A, "sdfr_splat_weights", primitive, K, Kinv, p_cam, n_cam, p_cam /* attr unused */, uv, znorm, nullptr, nullptr, B, cap,
                       cnt, W, H, diam, depth_constant);
    if (rc) return rc;
    SDFR_REQUIRE(aux && weights, "sdfr_splat_weights: NULL argument");
    rc = set_clamp(A, "sdfr_splat_weights_clamp", primitive, clamp_alt, clamp_constant);
    if (rc) return rc;
    if (B == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int rows = cap + (bg_logit ? 1 : 0);
    if (cap > 0) {
        const dim3 g(sdfr_cdiv(cap, 4), B);
        if (clamp_alt) switch (primitive) {
            case 0: hipLaunchKernelGGL((sdfr_splat_weights_kernel<0, true>), g, dim3(256), 0, s, A, aux, weights, rows); break;
            case 1: hipLaunchKernelGGL((sdfr_splat_weights_kernel<1, true>), g, dim3(256), 0, s, A, aux, weights, rows); break;
            default: hipLaunchKernelGGL((sdfr_splat_weights_kernel<2, true>), g, dim3(256), 0, s, A, aux, weights, rows); break;
        }
        else switch (primitive) {
            case 0: hipLaunchKernelGGL(sdfr_splat_weights_kernel<0>, g, dim3(256), 0, s, A, aux, weights, rows); break;
            case 1: hipLaunchKernelGGL(sdfr_splat_weights_kernel<1>, g, dim3(256), 0, s, A, aux, weights, rows); break;
            default: hipLaunchKernelGGL(sdfr_splat_weights_kernel<2>, g, dim3(256), 0, s, A, aux, weights, rows); break;
        }
    }
    if (bg_logit) hipLaunchKernelGGL(sdfr_splat_bgweights_kernel, dim3(sdfr_cdiv(W * H, 256), B), dim3(256), 0, s, aux, bg_logit, weights, rows, W * H);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_splat_weights_backward(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam,
                                           const float* uv, const float* znorm, int has_bg_row, int B, int cap, const int32_t* cnt, int W,
                                           int H, float diam, float depth_constant, const float* aux, const float* g_weights,
                                           const float* wsum, float* g_p_cam, float* g_n_cam, void* stream) {
    return sdfr_splat_weights_backward_clamp(primitive, K, Kinv, p_cam, n_cam, uv, znorm, has_bg_row, B, cap, cnt, W, H, diam, depth_constant, 0,
                                             primitive == 1 ? 3.f : 5.f, aux, g_weights, wsum, g_p_cam, g_n_cam, stream);
}

extern "C" int sdfr_splat_weights_backward_clamp(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam,
                                                 const float* uv, const float* znorm, int has_bg_row, int B, int cap, const int32_t* cnt, int W,
                                                 int H, float diam, float depth_constant, int clamp_alt, float clamp_constant, const float* aux,
                                                 const float* g_weights, const float* wsum, float* g_p_cam, float* g_n_cam, void* stream) {
    SplatArgs A;
    int rc = fill_args(A, "sdfr_splat_weights_backward", primitive, K, Kinv, p_cam, n_cam, p_cam /* attr unused */, uv, znorm, nullptr, nullptr,
                       B, cap, cnt, W, H, diam, depth_constant);
    if (rc) return rc;
    SDFR_REQUIRE(aux && g_weights && wsum && g_p_cam && g_n_cam, "sdfr_splat_weights_backward: NULL argument");
    rc = set_clamp(A, "sdfr_splat_weights_backward_clamp", primitive, clamp_alt, clamp_constant);
    if (rc) return rc;
    if (B == 0 || cap == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(sdfr_cdiv(cap, 4), B);
    const int rows = cap + (has_bg_row ? 1 : 0);
    const float* z = nullptr;
    float* zo = nullptr;
    if (clamp_alt) switch (primitive) {
        case 0: hipLaunchKernelGGL((sdfr_splat_bwd_kernel<0, true, true>), g, dim3(256), 0, s, A, aux, z, z, z, z, z, z, z, z, g_p_cam, g_n_cam, zo,
                                   g_weights, wsum, rows); break;
        case 1: hipLaunchKernelGGL((sdfr_splat_bwd_kernel<1, true, true>), g, dim3(256), 0, s, A, aux, z, z, z, z, z, z, z, z, g_p_cam, g_n_cam, zo,
                                   g_weights, wsum, rows); break;
        default: hipLaunchKernelGGL((sdfr_splat_bwd_kernel<2, true, true>), g, dim3(256), 0, s, A, aux, z, z, z, z, z, z, z, z, g_p_cam, g_n_cam, zo,
                                    g_weights, wsum, rows); break;
    }
    else switch (primitive) {
        case 0: hipLaunchKernelGGL((sdfr_splat_bwd_kernel<0, true>), g, dim3(256), 0, s, A, aux, z, z, z, z, z, z, z, z, g_p_cam, g_n_cam, zo,
                                   g_weights, wsum, rows); break;
        case 1: hipLaunchKernelGGL((sdfr_splat_bwd_kernel<1, true>), g, dim3(256), 0, s, A, aux, z, z, z, z, z, z, z, z, g_p_cam, g_n_cam, zo,
                                   g_weights, wsum, rows); break;
        default: hipLaunchKernelGGL((sdfr_splat_bwd_kernel<2, true>), g, dim3(256), 0, s, A, aux, z, z, z, z, z, z, z, z, g_p_cam, g_n_cam, zo,
                                    g_weights, wsum, rows); break;
    }
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
