// Object -> camera projection of the surfels, front-face filter, and the backward to (points, normals, pose).
//
// Replaces project_in_2D with rot='dcm' (reference sdfrenderer/renderer/projection.py:7-101): RT = pose[:3] (:34-41),
// n_c = R n (:49), NOCS colours c = p * (-1,1,1) (:53-55), p_c = RT [p;1] (:58), front-face mask n_c . p_c < 0 with an
// order-preserving compaction (:61-70; here wave ballots + a running offset, one workgroup per crop), pinhole
// projection K p_c / (z + eps) clamped to [-1,res] (:88-93).
// Compiled with -ffp-contract=off; fused multiply-adds only where written explicitly.
#include "sdfr_common.h"
#include "splat_bbox.h"
#include <float.h>

#define PROJ_THREADS 1024

// SURF: the surfels are produced here as well -- iso-surface projection of the band rows (Grid3D.get_surface_points, grid.py:57-67, the
// arithmetic of sdfr_surface_project) -- and written to points / normals; their conservative disc screen boxes (what
// sdfr_splat_forward would compute in a launch of its own) go to S.bbox.  One launch instead of three for the batched path.
struct SurfArgs {
    const float* xyz; int xyz_stride; const float* sdf; int64_t G; const int32_t* idx; const float* J; int Jstride, Joff;
    float* points_w; float* normals_w; int4* bbox; float diam;
    int32_t* bins;            // per-crop tile lists behind the boxes (splat_bbox.h), or NULL
    const int32_t* wh;        // ragged extents (r04): int32[B][2] = (W_b, H_b) per crop on the device, or NULL (every crop res_x x res_y)
    int64_t bin_stride;       // words per crop of the tile-list workspace
};

template <bool SURF>
__global__ __launch_bounds__(PROJ_THREADS) void sdfr_project_dcm_kernel(
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ points,
    const float* __restrict__ normals, const float* __restrict__ colors, int cap, const int32_t* __restrict__ cnt,
    int output_nocs, float res_x, float res_y, float* __restrict__ p_cam, float* __restrict__ n_cam, float* __restrict__ col,
    float* __restrict__ uv, int32_t* __restrict__ fidx, int32_t* __restrict__ fcnt, float* __restrict__ xyzf,
    int32_t* __restrict__ fslot, const SurfArgs S) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int count = sdfr_count(cnt, b, cap);
    if (SURF && S.wh) {                                                                    // this crop's own image size
        int w = S.wh[2 * b], h = S.wh[2 * b + 1];
        // an extent outside the caller's contract (include/sdfr.h) -- or with more 8x8 tiles than the tile-list layout holds -- is an EMPTY crop
        const int64_t tcap = S.bins ? S.bin_stride - 2 - (int64_t)SPL_LM * cap : (int64_t)1 << 40;
        if (w < 1 || h < 1 || (int64_t)((w + 7) >> 3) * ((h + 7) >> 3) > tcap) { w = 0; h = 0; }
        res_x = (float)w; res_y = (float)h;
    }
    const float* P = pose + (int64_t)b * 16;
    const float* Kb = K + (int64_t)b * 9;
    const float r00 = P[0], r01 = P[1], r02 = P[2], t0 = P[3];
    const float r10 = P[4], r11 = P[5], r12 = P[6], t1 = P[7];
    const float r20 = P[8], r21 = P[9], r22 = P[10], t2 = P[11];
    __shared__ int wc[PROJ_THREADS / 64 + 1];
    __shared__ int s_base;
    __shared__ int tile_cnt[SURF ? SPL_BIN_MAX_TILES : 1];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int s0 = 0; s0 < count; s0 += PROJ_THREADS) {
        const int s = s0 + tid;
        bool front = false;
        float fx = 0.f, fy = 0.f, fz = 0.f;
        if (s < count) {
            const int64_t e = ((int64_t)b * cap + s) * 3;
            float x, y, z, nx, ny, nz;
            if (SURF) {
                const int64_t e1 = (int64_t)b * cap + s;
                const int64_t r = (int64_t)b * S.G + S.idx[e1];
                const float* xg = S.xyz + r * S.xyz_stride;
                const float* n = S.J + e1 * S.Jstride + S.Joff;
                const float jx = n[0], jy = n[1], jz = n[2];
                const float nrm = sqrtf(jx * jx + jy * jy + jz * jz);          // grid.py:57
                nx = jx / nrm; ny = jy / nrm; nz = jz / nrm;                   // grid.py:58
                const float sd = S.sdf[r];
                x = xg[0] - sd * nx; y = xg[1] - sd * ny; z = xg[2] - sd * nz;   // grid.py:61
                S.points_w[e] = x; S.points_w[e + 1] = y; S.points_w[e + 2] = z;
                S.normals_w[e] = nx; S.normals_w[e + 1] = ny; S.normals_w[e + 2] = nz;
            } else {
                x = points[e]; y = points[e + 1]; z = points[e + 2];
                nx = normals[e]; ny = normals[e + 1]; nz = normals[e + 2];
            }
            // p_c = RT [p;1]  (:58)   n_c = R n  (:49)
            const float pcx = fmaf(r02, z, fmaf(r01, y, r00 * x)) + t0;
            const float pcy = fmaf(r12, z, fmaf(r11, y, r10 * x)) + t1;
            const float pcz = fmaf(r22, z, fmaf(r21, y, r20 * x)) + t2;
            const float ncx = fmaf(r02, nz, fmaf(r01, ny, r00 * nx));
            const float ncy = fmaf(r12, nz, fmaf(r11, ny, r10 * nx));
            const float ncz = fmaf(r22, nz, fmaf(r21, ny, r20 * nx));
            p_cam[e] = pcx; p_cam[e + 1] = pcy; p_cam[e + 2] = pcz;
            n_cam[e] = ncx; n_cam[e + 1] = ncy; n_cam[e + 2] = ncz;
            fx = pcx; fy = pcy; fz = pcz;
            if (SURF && S.bbox) {
                int x0, y0, x1, y1;
                const bool ok = disc_bbox(Kb, pcx, pcy, pcz, S.diam, (int)res_x, (int)res_y, x0, y0, x1, y1);
                S.bbox[(int64_t)b * cap + s] = ok ? make_int4(x0, y0, x1, y1) : make_int4(1, 1, 0, 0);
            }
            if (output_nocs) {                      // :53-55 (2: quat path, no flip :147-149); +4: the compositing map (c+1)/2 applied here
                float c0 = ((output_nocs & 3) == 2) ? x : -x, c1 = y, c2 = z;
                if (output_nocs & 4) { c0 = (c0 + 1.f) * 0.5f; c1 = (c1 + 1.f) * 0.5f; c2 = (c2 + 1.f) * 0.5f; }      // rasterer.py:113-114
                col[e] = c0; col[e + 1] = c1; col[e + 2] = c2;
            } else { col[e] = colors[e]; col[e + 1] = colors[e + 1]; col[e + 2] = colors[e + 2]; }
            const float dot = ncx * pcx + ncy * pcy + ncz * pcz;                          // :62
            front = dot < 0.f;
            if (uv) {
                const float hx = fmaf(Kb[2], pcz, fmaf(Kb[1], pcy, Kb[0] * pcx));
                const float hy = fmaf(Kb[5], pcz, fmaf(Kb[4], pcy, Kb[3] * pcx));
                const float hz = fmaf(Kb[8], pcz, fmaf(Kb[7], pcy, Kb[6] * pcx));
                const float den = hz + FLT_EPSILON;                                        // :89
                const int64_t e2 = ((int64_t)b * cap + s) * 2;
                uv[e2] = fminf(fmaxf(hx / den, -1.f), res_x);                              // :92
                uv[e2 + 1] = fminf(fmaxf(hy / den, -1.f), res_y);                          // :93
            }
        }
        if (fidx) {
            const unsigned long long bal = __ballot(front);
            if (lane == 0) wc[wv] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wv; ++w) woff += wc[w];
            const int base = s_base;
            const int slot = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
            if (front) {
                fidx[(int64_t)b * cap + slot] = s;
                if (xyzf) { const int64_t f = ((int64_t)b * cap + slot) * 3; xyzf[f] = fx; xyzf[f + 1] = fy; xyzf[f + 2] = fz; }   // points['xyzf'], rasterer.py:151
            }
            if (fslot && s < count) fslot[(int64_t)b * cap + s] = front ? slot : -1;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < PROJ_THREADS / 64; ++w) tot += wc[w];
                s_base = base + tot;
            }
            __syncthreads();
        }
    }
    if (fcnt && tid == 0) fcnt[b] = s_base;
    if (SURF && S.bbox && S.bins) {
        // the crop's boxes (written above by this workgroup) -> per-tile surfel lists for the splat kernel
        __threadfence_block();
        __syncthreads();
        sdfr_bin_boxes<PROJ_THREADS>(S.bbox + (int64_t)b * cap, count, (int)res_x, (int)res_y, cap, S.bins + (int64_t)b * S.bin_stride, tile_cnt, wc);
    }
}

extern "C" int sdfr_project_dcm(const float* pose, const float* K, const float* points, const float* normals,
                                const float* colors, int B, int cap, const int32_t* cnt, int output_nocs, int res_x, int res_y,
                                float* p_cam, float* n_cam, float* col, float* uv, int32_t* fidx, int32_t* fcnt, float* xyzf,
                                int32_t* fslot, void* stream) {
    SDFR_REQUIRE(pose && K && points && normals && p_cam && n_cam && col, "sdfr_project_dcm: NULL argument");
    SDFR_REQUIRE(output_nocs || colors, "sdfr_project_dcm: colors required when output_nocs == 0");
    SDFR_REQUIRE((fidx == nullptr) == (fcnt == nullptr), "sdfr_project_dcm: fidx and fcnt must be given together");
    SDFR_REQUIRE(fidx || (!xyzf && !fslot), "sdfr_project_dcm: xyzf / fslot need fidx and fcnt");
    SDFR_REQUIRE(output_nocs >= 0 && output_nocs <= 6 && output_nocs != 3 && output_nocs != 4, "sdfr_project_dcm: output_nocs %d unknown",
                 output_nocs);
    if (B <= 0) return SDFR_OK;
    SurfArgs S = {};
    hipLaunchKernelGGL(sdfr_project_dcm_kernel<false>, dim3(B), dim3(PROJ_THREADS), 0, (hipStream_t)stream, pose, K, points, normals,
                       colors, cap, cnt, output_nocs, (float)res_x, (float)res_y, p_cam, n_cam, col, uv, fidx, fcnt, xyzf, fslot, S);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// Band rows -> surfels -> camera frame -> front-face list -> screen boxes in ONE launch (batched path): sdfr_surface_project +
// sdfr_project_dcm (NOCS colour modes) + the box pass of sdfr_splat_forward (disc primitive), same arithmetic, same outputs.
static int surfels_forward_impl(const char* who, const float* xyz, int xyz_stride, const float* sdf, int64_t G, const int32_t* idx, const float* J,
                                int Jstride, int Joff, const float* pose, const float* K, int B, int cap, const int32_t* cnt, int output_nocs,
                                int res_x, int res_y, const int32_t* wh, int tiles_cap, float diam, float* points, float* normals, float* p_cam,
                                float* n_cam, float* col, int32_t* fidx, int32_t* fcnt, float* xyzf, int32_t* fslot, int32_t* bbox,
                                void* stream) {
    SDFR_REQUIRE(xyz && sdf && idx && J && pose && K && points && normals && p_cam && n_cam && col, "%s: NULL argument", who);
    const bool no_bins = (output_nocs & 8) == 0;          // | 8: bbox is the large workspace, build the tile lists too (SDFR_PRIM_BINS)
    output_nocs &= ~8;
    SDFR_REQUIRE(output_nocs == 1 || output_nocs == 2 || output_nocs == 5 || output_nocs == 6, "%s: NOCS colour modes only", who);
    SDFR_REQUIRE((fidx == nullptr) == (fcnt == nullptr), "%s: fidx and fcnt must be given together", who);
    SDFR_REQUIRE(fidx || (!xyzf && !fslot), "%s: xyzf / fslot need fidx and fcnt", who);
    SDFR_REQUIRE(!wh || no_bins || (tiles_cap > 0 && tiles_cap <= SPL_BIN_MAX_TILES), "%s: tile lists need 0 < tiles_cap <= %d", who, SPL_BIN_MAX_TILES);
    if (B <= 0) return SDFR_OK;
    SurfArgs S = {xyz, xyz_stride, sdf, G, idx, J, Jstride, Joff, points, normals, reinterpret_cast<int4*>(bbox), diam,
                  (bbox && !no_bins) ? bbox + (int64_t)B * cap * 4 : nullptr, wh,
                  wh ? (int64_t)tiles_cap + 2 + (int64_t)SPL_LM * cap : sdfr_splat_bin_stride(cap, res_x, res_y)};
    hipLaunchKernelGGL(sdfr_project_dcm_kernel<true>, dim3(B), dim3(PROJ_THREADS), 0, (hipStream_t)stream, pose, K, nullptr, nullptr,
                       nullptr, cap, cnt, output_nocs, (float)res_x, (float)res_y, p_cam, n_cam, col, nullptr, fidx, fcnt, xyzf, fslot, S);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_surfels_forward(const float* xyz, int xyz_stride, const float* sdf, int64_t G, const int32_t* idx, const float* J,
                                    int Jstride, int Joff, const float* pose, const float* K, int B, int cap, const int32_t* cnt,
                                    int output_nocs, int res_x, int res_y, float diam, float* points, float* normals, float* p_cam,
                                    float* n_cam, float* col, int32_t* fidx, int32_t* fcnt, float* xyzf, int32_t* fslot,
                                    int32_t* bbox, void* stream) {
    return surfels_forward_impl("sdfr_surfels_forward", xyz, xyz_stride, sdf, G, idx, J, Jstride, Joff, pose, K, B, cap, cnt, output_nocs, res_x,
                                res_y, nullptr, 0, diam, points, normals, p_cam, n_cam, col, fidx, fcnt, xyzf, fslot, bbox, stream);
}

// ragged extents: crop b projects into its own W_b x H_b image (wh int32[B][2] on the device); the tile-list workspace is laid out for
// tiles_cap tiles per crop (sdfr_splat_ws_words_r)
extern "C" int sdfr_surfels_forward_r(const float* xyz, int xyz_stride, const float* sdf, int64_t G, const int32_t* idx, const float* J,
                                      int Jstride, int Joff, const float* pose, const float* K, int B, int cap, const int32_t* cnt,
                                      int output_nocs, const int32_t* wh, int tiles_cap, float diam, float* points, float* normals,
                                      float* p_cam, float* n_cam, float* col, int32_t* fidx, int32_t* fcnt, float* xyzf, int32_t* fslot,
                                      int32_t* bbox, void* stream) {
    SDFR_REQUIRE(wh, "sdfr_surfels_forward_r: NULL extents");
    return surfels_forward_impl("sdfr_surfels_forward_r", xyz, xyz_stride, sdf, G, idx, J, Jstride, Joff, pose, K, B, cap, cnt, output_nocs, 1, 1,
                                wh, tiles_cap, diam, points, normals, p_cam, n_cam, col, fidx, fcnt, xyzf, fslot, bbox, stream);
}

// ---- backward ----------------------------------------------------------------------------------------------------
// g_points = R^T g_pc (+ g_col * (-1,1,1) for NOCS), g_normals = R^T g_nc, g_pose[:3,:3] = sum g_pc p^T + g_nc n^T,
// g_pose[:3,3] = sum g_pc.  One workgroup per crop; the 12 pose sums use a fixed-order tree (deterministic).

__global__ __launch_bounds__(PROJ_THREADS) void sdfr_project_dcm_bwd_kernel(
    const float* __restrict__ pose, const float* __restrict__ points, const float* __restrict__ normals,
    const float* __restrict__ g_pc, const float* __restrict__ g_nc, const float* __restrict__ g_col, int cap,
    const int32_t* __restrict__ cnt, int output_nocs, float* __restrict__ g_points, float* __restrict__ g_normals,
    float* __restrict__ g_colors, float* __restrict__ g_pose, const float* __restrict__ g_xyzf, const int32_t* __restrict__ fslot) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int count = sdfr_count(cnt, b, cap);
    const float* P = pose + (int64_t)b * 16;
    const float r00 = P[0], r01 = P[1], r02 = P[2];
    const float r10 = P[4], r11 = P[5], r12 = P[6];
    const float r20 = P[8], r21 = P[9], r22 = P[10];
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int s = tid; s < count; s += PROJ_THREADS) {
        const int64_t e = ((int64_t)b * cap + s) * 3;
        const float x = points[e], y = points[e + 1], z = points[e + 2];
        const float nx = normals[e], ny = normals[e + 1], nz = normals[e + 2];
        float ax = g_pc ? g_pc[e] : 0.f, ay = g_pc ? g_pc[e + 1] : 0.f, az = g_pc ? g_pc[e + 2] : 0.f;
        if (g_xyzf) {                                   // gradient arriving through points['xyzf'] (the 3-D loss), added to the renderer's
            const int fs = fslot[(int64_t)b * cap + s];
            if (fs >= 0) { const int64_t f = ((int64_t)b * cap + fs) * 3; ax += g_xyzf[f]; ay += g_xyzf[f + 1]; az += g_xyzf[f + 2]; }
        }
        const float bx = g_nc ? g_nc[e] : 0.f, by = g_nc ? g_nc[e + 1] : 0.f, bz = g_nc ? g_nc[e + 2] : 0.f;
        float gx = r00 * ax + r10 * ay + r20 * az;
        float gy = r01 * ax + r11 * ay + r21 * az;
        float gz = r02 * ax + r12 * ay + r22 * az;
        if (g_col) {
            if (output_nocs) {
                float c0 = g_col[e], c1 = g_col[e + 1], c2 = g_col[e + 2];
                if (output_nocs & 4) { c0 *= 0.5f; c1 *= 0.5f; c2 *= 0.5f; }
                gx += ((output_nocs & 3) == 2) ? c0 : -c0; gy += c1; gz += c2;
            }
            else if (g_colors) { g_colors[e] = g_col[e]; g_colors[e + 1] = g_col[e + 1]; g_colors[e + 2] = g_col[e + 2]; }
        } else if (!output_nocs && g_colors) { g_colors[e] = 0.f; g_colors[e + 1] = 0.f; g_colors[e + 2] = 0.f; }
        g_points[e] = gx; g_points[e + 1] = gy; g_points[e + 2] = gz;
        g_normals[e] = r00 * bx + r10 * by + r20 * bz;
        g_normals[e + 1] = r01 * bx + r11 * by + r21 * bz;
        g_normals[e + 2] = r02 * bx + r12 * by + r22 * bz;
        acc[0] += ax * x + bx * nx; acc[1] += ax * y + bx * ny; acc[2] += ax * z + bx * nz; acc[3] += ax;
        acc[4] += ay * x + by * nx; acc[5] += ay * y + by * ny; acc[6] += ay * z + by * nz; acc[7] += ay;
        acc[8] += az * x + bz * nx; acc[9] += az * y + bz * ny; acc[10] += az * z + bz * nz; acc[11] += az;
    }
    __shared__ float red[12][PROJ_THREADS / 64];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) red[i][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 16) {
        float v = 0.f;
        if (tid < 12)
            for (int w = 0; w < PROJ_THREADS / 64; ++w) v += red[tid][w];
        g_pose[(int64_t)b * 16 + tid] = v;
    }
}

extern "C" int sdfr_project_dcm_bwd(const float* pose, const float* points, const float* normals, const float* g_p_cam,
                                    const float* g_n_cam, const float* g_col, int B, int cap, const int32_t* cnt, int output_nocs,
                                    float* g_points, float* g_normals, float* g_colors, float* g_pose, const float* g_xyzf,
                                    const int32_t* fslot, void* stream) {
    SDFR_REQUIRE(pose && points && normals && g_points && g_normals && g_pose, "sdfr_project_dcm_bwd: NULL argument");
    SDFR_REQUIRE(!g_xyzf || fslot, "sdfr_project_dcm_bwd: g_xyzf needs the fslot array sdfr_project_dcm wrote");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_project_dcm_bwd_kernel, dim3(B), dim3(PROJ_THREADS), 0, (hipStream_t)stream, pose, points, normals,
                       g_p_cam, g_n_cam, g_col, cap, cnt, output_nocs, g_points, g_normals, g_colors, g_pose, g_xyzf, fslot);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
