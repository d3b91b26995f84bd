// Parameter glue of the batched refinement step: optimizer parameters <-> renderer inputs, entirely on the device.
//
// Restates, for B crops at once, the host-side tensor algebra the reference's optimizer executes per iteration with dozens of
// tiny ATen launches (pipelines/optimizer.py:86-100): pose = [R_y(yaw) | t] with row 1 of the rotation negated (:87-90,
// utils/refinement.py:108-125), latent_ = F.normalize(latent, p=2, dim=0) (:96), inputs = cat(latent_.expand(G,-1), grid.points)
// (:99-100); and the matching backward.  Compiled with -ffp-contract=off.
#include "sdfr_common.h"
#include "solver.h"

__global__ __launch_bounds__(256) void sdfr_params_forward_kernel(const float* __restrict__ yaw, const float* __restrict__ trans,
                                                                 const float* __restrict__ latent, int L,
                                                                 const float* __restrict__ grid, int64_t G,
                                                                 float* __restrict__ inputs, float* __restrict__ pose,
                                                                 float* __restrict__ latnorm) {
    const int b = blockIdx.y;
    const int NI = L + 3;
    // ||latent||_2, max(., 1e-12) as F.normalize does
    float ss = 0.f;
    for (int c = 0; c < L; ++c) ss += latent[b * L + c] * latent[b * L + c];
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        latnorm[b] = nrm;
        const float c = cosf(yaw[b]), s = sinf(yaw[b]);
        float* P = pose + (int64_t)b * 16;
        P[0] = c;   P[1] = 0.f;  P[2] = s;   P[3] = trans[b * 3 + 0];
        P[4] = -0.f; P[5] = -1.f; P[6] = -0.f; P[7] = trans[b * 3 + 1];
        P[8] = -s;  P[9] = 0.f;  P[10] = c;  P[11] = trans[b * 3 + 2];
        P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
    }
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= G || !inputs) return;
    float* row = inputs + ((int64_t)b * G + g) * NI;
    for (int c = 0; c < L; ++c) row[c] = latent[b * L + c] / nrm;
    row[L] = grid[g * 3]; row[L + 1] = grid[g * 3 + 1]; row[L + 2] = grid[g * 3 + 2];
}

extern "C" int sdfr_params_forward(const float* yaw, const float* trans, const float* latent, int L, const float* grid, int64_t G,
                                   int B, float* inputs, float* pose, float* latnorm, void* stream) {
    // inputs == NULL: pose and latent norm only (pose-only refinement: the decoder rows of a frozen shape stay as they are)
    SDFR_REQUIRE(yaw && trans && latent && (grid || !inputs) && pose && latnorm, "sdfr_params_forward: NULL argument");
    SDFR_REQUIRE(L >= 0 && L <= 1024 && G > 0, "sdfr_params_forward: bad size");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_params_forward_kernel, dim3(inputs ? sdfr_cdiv(G, 256) : 1, B), dim3(inputs ? 256 : 64), 0, (hipStream_t)stream, yaw, trans, latent,
                       L, grid, G, inputs, pose, latnorm);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// sdfr_params_forward + sdfr_prefilter_plan (csrc/surface.hip) in ONE launch (r06): the plan compares the crop's normalised latent -- the
// latent columns params_forward writes, latent / max(||latent||, 1e-12): recomputed here by the same division -- with the latent of the crop's
// last full-grid pass, so thread 0 of the crop's first block decides it while the other blocks write the input rows.  Same outputs, bit for bit.
__global__ __launch_bounds__(256) void sdfr_params_plan_kernel(const float* __restrict__ yaw, const float* __restrict__ trans,
                                                              const float* __restrict__ latent, int L, const float* __restrict__ grid, int64_t G,
                                                              float* __restrict__ inputs, float* __restrict__ pose, float* __restrict__ latnorm,
                                                              float lip, const float* __restrict__ margin, const float* __restrict__ max_dev,
                                                              float* __restrict__ lat_ref, int32_t* __restrict__ age, int max_reuse,
                                                              int32_t* __restrict__ reuse, int32_t* __restrict__ n_full) {
    const int b = blockIdx.y;
    const int NI = L + 3;
    float ss = 0.f;
    for (int c = 0; c < L; ++c) ss += latent[b * L + c] * latent[b * L + c];
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        latnorm[b] = nrm;
        const float c = cosf(yaw[b]), s = sinf(yaw[b]);
        float* P = pose + (int64_t)b * 16;
        P[0] = c;   P[1] = 0.f;  P[2] = s;   P[3] = trans[b * 3 + 0];
        P[4] = -0.f; P[5] = -1.f; P[6] = -0.f; P[7] = trans[b * 3 + 1];
        P[8] = -s;  P[9] = 0.f;  P[10] = c;  P[11] = trans[b * 3 + 2];
        P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
        // the plan (sdfr_prefilter_plan_kernel, statement for statement; z[c] = the row value latent / nrm)
        float d2 = 0.f;
        for (int c2 = 0; c2 < L; ++c2) { const float d = latent[b * L + c2] / nrm - lat_ref[b * L + c2]; d2 += d * d; }
        const bool ok = age[b] > 0 && age[b] <= max_reuse && lip * sqrtf(d2) <= 0.25f * margin[b] && max_dev[b] <= 0.5f * margin[b];
        reuse[b] = ok ? 1 : 0;
        if (ok) age[b] += 1;
        else {
            age[b] = 1;
            for (int c2 = 0; c2 < L; ++c2) lat_ref[b * L + c2] = latent[b * L + c2] / nrm;
            if (n_full) n_full[b] += 1;
        }
    }
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    float* row = inputs + ((int64_t)b * G + g) * NI;
    for (int c = 0; c < L; ++c) row[c] = latent[b * L + c] / nrm;
    row[L] = grid[g * 3]; row[L + 1] = grid[g * 3 + 1]; row[L + 2] = grid[g * 3 + 2];
}

extern "C" int sdfr_params_plan(const float* yaw, const float* trans, const float* latent, int L, const float* grid, int64_t G, int B, float* inputs,
                                float* pose, float* latnorm, float lip, const float* margin, const float* max_dev, float* lat_ref, int32_t* age,
                                int max_reuse, int32_t* reuse, int32_t* n_full, void* stream) {
    SDFR_REQUIRE(yaw && trans && latent && grid && inputs && pose && latnorm && margin && max_dev && lat_ref && age && reuse,
                 "sdfr_params_plan: NULL argument");
    SDFR_REQUIRE(L >= 0 && L <= 1024 && G > 0, "sdfr_params_plan: bad size");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_params_plan_kernel, dim3(sdfr_cdiv(G, 256), B), dim3(256), 0, (hipStream_t)stream, yaw, trans, latent, L, grid, G, inputs,
                       pose, latnorm, lip, margin, max_dev, lat_ref, age, max_reuse, reuse, n_full);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// g_latn[b][c] = sum_s g_sdf_s * J[b][s][c],  g_sdf_s = -(g_points_s . n_hat_s)      (grid.py:61 backward, then the decoder's
// input gradient summed over the expanded latent rows).  One workgroup per crop, fixed-order tree (deterministic).
#define LAT_THREADS 1024
#define LAT_MAXL 8
__global__ __launch_bounds__(LAT_THREADS) void sdfr_surface_latent_grad_kernel(const float* __restrict__ g_points,
                                                                              const float* __restrict__ g_nocs,
                                                                              const float* __restrict__ normals,
                                                                              const float* __restrict__ J, int NI, int L, int cap,
                                                                              const int32_t* __restrict__ cnt,
                                                                              float* __restrict__ g_latn) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int count = sdfr_count(cnt, b, cap);
    __shared__ float red[LAT_THREADS / 64];
    for (int c0 = 0; c0 < L; c0 += LAT_MAXL) {
        float acc[LAT_MAXL];
#pragma unroll
        for (int i = 0; i < LAT_MAXL; ++i) acc[i] = 0.f;
        for (int s = tid; s < count; s += LAT_THREADS) {
            const int64_t e = (int64_t)b * cap + s;
            float gx = g_points[e * 3], gy = g_points[e * 3 + 1], gz = g_points[e * 3 + 2];
            if (g_nocs) { gx += g_nocs[e * 3] / 2.f; gy += g_nocs[e * 3 + 1] / 2.f; gz += g_nocs[e * 3 + 2] / 2.f; }
            const float gs = -(gx * normals[e * 3] + gy * normals[e * 3 + 1] + gz * normals[e * 3 + 2]);
#pragma unroll
            for (int i = 0; i < LAT_MAXL; ++i)
                if (c0 + i < L) acc[i] += gs * J[e * NI + c0 + i];
        }
#pragma unroll
        for (int i = 0; i < LAT_MAXL; ++i) {
            if (c0 + i >= L) break;
            float v = acc[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = v;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (int w = 0; w < LAT_THREADS / 64; ++w) t += red[w];
                g_latn[b * L + c0 + i] = t;
            }
        }
    }
}

extern "C" int sdfr_surface_latent_grad(const float* g_points, const float* g_nocs, const float* normals, const float* J, int n_inputs,
                                        int L, int B, int cap, const int32_t* cnt, float* g_latn, void* stream) {
    SDFR_REQUIRE(g_points && normals && J && g_latn, "sdfr_surface_latent_grad: NULL argument");
    SDFR_REQUIRE(L >= 0 && L <= n_inputs, "sdfr_surface_latent_grad: bad latent size");
    if (B <= 0 || L == 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_surface_latent_grad_kernel, dim3(B), dim3(LAT_THREADS), 0, (hipStream_t)stream, g_points, g_nocs, normals,
                       J, n_inputs, L, cap, cnt, g_latn);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// g_yaw, g_trans from g_pose (optimizer.py:87-90); g_latent from g_latn through F.normalize (optimizer.py:96)
__global__ __launch_bounds__(64) void sdfr_params_backward_kernel(const float* __restrict__ yaw, const float* __restrict__ latent, int L,
                                                                 const float* __restrict__ latnorm, const float* __restrict__ g_pose,
                                                                 const float* __restrict__ g_latn, int B, float* __restrict__ g_yaw,
                                                                 float* __restrict__ g_trans, float* __restrict__ g_latent) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* g = g_pose + (int64_t)b * 16;
    const float c = cosf(yaw[b]), s = sinf(yaw[b]);
    // R = [[c,0,s],[0,-1,0],[-s,0,c]]  ->  dR/dyaw = [[-s,0,c],[0,0,0],[-c,0,-s]]
    g_yaw[b] = (-s) * g[0] + c * g[2] + (-c) * g[8] + (-s) * g[10];
    g_trans[b * 3] = g[3]; g_trans[b * 3 + 1] = g[7]; g_trans[b * 3 + 2] = g[11];
    const float nrm = latnorm[b];
    float dot = 0.f;
    for (int i = 0; i < L; ++i) dot += (latent[b * L + i] / nrm) * g_latn[b * L + i];
    for (int i = 0; i < L; ++i) g_latent[b * L + i] = (g_latn[b * L + i] - (latent[b * L + i] / nrm) * dot) / nrm;
}

extern "C" int sdfr_params_backward(const float* yaw, const float* latent, int L, const float* latnorm, const float* g_pose,
                                    const float* g_latn, int B, float* g_yaw, float* g_trans, float* g_latent, void* stream) {
    SDFR_REQUIRE(yaw && latent && latnorm && g_pose && g_latn && g_yaw && g_trans && g_latent, "sdfr_params_backward: NULL argument");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_params_backward_kernel, dim3(sdfr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, yaw, latent, L, latnorm,
                       g_pose, g_latn, B, g_yaw, g_trans, g_latent);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// dst[b][idx[b][j]][:] += src[b][j][:] for j < cnt[b]  (gradient of the front-facing selection points_3d_filt, projection.py:64-70)
__global__ __launch_bounds__(256) void sdfr_scatter_add_rows3_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                    const int32_t* __restrict__ idx, int cap,
                                                                    const int32_t* __restrict__ cnt) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= sdfr_count(cnt, b, cap)) return;
    const int64_t d = ((int64_t)b * cap + idx[(int64_t)b * cap + j]) * 3, s = ((int64_t)b * cap + j) * 3;
    dst[d] += src[s]; dst[d + 1] += src[s + 1]; dst[d + 2] += src[s + 2];
}

extern "C" int sdfr_scatter_add_rows3(float* dst, const float* src, const int32_t* idx, int B, int cap, const int32_t* cnt,
                                      void* stream) {
    SDFR_REQUIRE(dst && src && idx, "sdfr_scatter_add_rows3: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_scatter_add_rows3_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, (hipStream_t)stream, dst, src, idx,
                       cap, cnt);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// out[b][j][:] = src[b][idx[b][j]][:] for j < cnt[b], zero beyond (points_3d_filt as a padded array)
__global__ __launch_bounds__(256) void sdfr_gather_rows3_kernel(float* __restrict__ out, const float* __restrict__ src,
                                                               const int32_t* __restrict__ idx, int cap,
                                                               const int32_t* __restrict__ cnt) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cap) return;
    const int64_t o = ((int64_t)b * cap + j) * 3;
    if (j < sdfr_count(cnt, b, cap)) {
        const int64_t s = ((int64_t)b * cap + idx[(int64_t)b * cap + j]) * 3;
        out[o] = src[s]; out[o + 1] = src[s + 1]; out[o + 2] = src[s + 2];
    } else {
        out[o] = 0.f; out[o + 1] = 0.f; out[o + 2] = 0.f;
    }
}

extern "C" int sdfr_gather_rows3(float* out, const float* src, const int32_t* idx, int B, int cap, const int32_t* cnt, void* stream) {
    SDFR_REQUIRE(out && src && idx, "sdfr_gather_rows3: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_gather_rows3_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, (hipStream_t)stream, out, src, idx, cap, cnt);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ---- fused backward tail of the batched step ----------------------------------------------------------------------------------
// sdfr_project_dcm_bwd + sdfr_surface_latent_grad + sdfr_params_backward in ONE launch (one workgroup per crop; latent sizes up to
// LAT_MAXL): per surfel the object-frame point gradient, its contraction with the normal (g_sdf) and with the Jacobian's latent columns,
// the 12 pose sums; fixed-order reductions exactly as in the three separate kernels (bit-identical results), then the parameter
// gradients.  Three launches of a few microseconds each become one.
#define PLB_THREADS 1024
#define PLB_UNROLL 3
// XYZF: a gradient arrives through the front-facing rows (g_xyzf / fslot); LAT: the latent columns of J are contracted (J != NULL).
// Absent optional inputs (g_pc, g_nc, g_col) are read from a valid dummy address and discarded, so that every load of a round is issued
// before the first wait (uniform NULL branches around single loads serialised them: one L2 round trip each, 13.5 us per launch);
// PLB_UNROLL rounds are loaded together.  The accumulation order per thread is unchanged (bit-identical results).
// SOLVE (r06): the gradient through xyzf arrives un-normalised with the per-crop factor kscale[2 b + 1] (sdfr_losses_fused: multiplied on load, the
// product the 3-D loss's finalize pass used to store), and the crop's solver step (sdfr_solver_step) follows its parameter gradients in the
// same thread -- the refinement iteration's last two launches in one.
struct SolveArgs {
    const float* kscale; float* params; const float* grads; const float* loss2d; const float* loss3d; const int32_t* npairs; float w2, w3;
    float* adam_m; float* adam_v; int32_t* adam_t; float lr_adam, lr_scale, lr_latent; int B; float* total; int32_t* stepped;
};
template <bool XYZF, bool LAT, bool SOLVE = false>
__global__ __launch_bounds__(PLB_THREADS) void sdfr_pose_latent_backward_kernel(
    const float* __restrict__ pose, const float* __restrict__ points, const float* __restrict__ normals, const float* __restrict__ g_pc,
    const float* __restrict__ g_nc, const float* __restrict__ g_col, int cap, const int32_t* __restrict__ cnt, int output_nocs,
    const float* __restrict__ g_xyzf, const int32_t* __restrict__ fslot, const float* __restrict__ J, int NI, int L,
    const float* __restrict__ yaw, const float* __restrict__ latent, const float* __restrict__ latnorm, float* __restrict__ g_points,
    float* __restrict__ g_pose, float* __restrict__ g_latn, float* g_yaw, float* g_trans,
    float* g_latent, const SolveArgs SA = SolveArgs()) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int count = sdfr_count(cnt, b, cap);
    const float* P = pose + (int64_t)b * 16;
    const float r00 = P[0], r01 = P[1], r02 = P[2];
    const float r10 = P[4], r11 = P[5], r12 = P[6];
    const float r20 = P[8], r21 = P[9], r22 = P[10];
    const float kx = SOLVE ? SA.kscale[2 * b + 1] : 1.f;
    float acc[12], lat[LAT_MAXL];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < LAT_MAXL; ++i) lat[i] = 0.f;
    const bool has_pc = g_pc != nullptr, has_nc = g_nc != nullptr, has_col = (g_col != nullptr) && output_nocs != 0;
    const float* q_pc = has_pc ? g_pc : points;
    const float* q_nc = has_nc ? g_nc : points;
    const float* q_col = has_col ? g_col : points;
    for (int s0 = tid; s0 < count; s0 += PLB_UNROLL * PLB_THREADS) {
        float3 pt[PLB_UNROLL], nm[PLB_UNROLL], va[PLB_UNROLL], vb[PLB_UNROLL], vc[PLB_UNROLL], vx[PLB_UNROLL];
        float jl[PLB_UNROLL][LAT_MAXL];
        bool ok[PLB_UNROLL], fx[PLB_UNROLL];
#pragma unroll
        for (int u = 0; u < PLB_UNROLL; ++u) {
            const int s = s0 + u * PLB_THREADS;
            ok[u] = s < count;
            const int64_t e1 = (int64_t)b * cap + (ok[u] ? s : s0);
            const int64_t e = e1 * 3;
            pt[u] = make_float3(points[e], points[e + 1], points[e + 2]);
            nm[u] = make_float3(normals[e], normals[e + 1], normals[e + 2]);
            va[u] = make_float3(q_pc[e], q_pc[e + 1], q_pc[e + 2]);
            vb[u] = make_float3(q_nc[e], q_nc[e + 1], q_nc[e + 2]);
            vc[u] = make_float3(q_col[e], q_col[e + 1], q_col[e + 2]);
            vx[u] = make_float3(0.f, 0.f, 0.f);
            fx[u] = false;
            if (XYZF) {
                const int fs = fslot[e1];
                fx[u] = fs >= 0;
                const int64_t f = ((int64_t)b * cap + (fx[u] ? fs : 0)) * 3;                 // back-facing rows read slot 0 and discard it
                if (SOLVE) vx[u] = make_float3(g_xyzf[f] * kx, g_xyzf[f + 1] * kx, g_xyzf[f + 2] * kx);
                else vx[u] = make_float3(g_xyzf[f], g_xyzf[f + 1], g_xyzf[f + 2]);
            }
            if (LAT) {
#pragma unroll
                for (int i = 0; i < LAT_MAXL; ++i) jl[u][i] = (i < L) ? J[e1 * NI + i] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < PLB_UNROLL; ++u) {
            if (!ok[u]) continue;
            const int64_t e = ((int64_t)b * cap + s0 + u * PLB_THREADS) * 3;
            const float x = pt[u].x, y = pt[u].y, z = pt[u].z;
            const float nx = nm[u].x, ny = nm[u].y, nz = nm[u].z;
            float ax = has_pc ? va[u].x : 0.f, ay = has_pc ? va[u].y : 0.f, az = has_pc ? va[u].z : 0.f;
            if (XYZF && fx[u]) { ax += vx[u].x; ay += vx[u].y; az += vx[u].z; }
            const float bx = has_nc ? vb[u].x : 0.f, by = has_nc ? vb[u].y : 0.f, bz = has_nc ? vb[u].z : 0.f;
            float gx = r00 * ax + r10 * ay + r20 * az;
            float gy = r01 * ax + r11 * ay + r21 * az;
            float gz = r02 * ax + r12 * ay + r22 * az;
            if (has_col) {
                float c0 = vc[u].x, c1 = vc[u].y, c2 = vc[u].z;
                if (output_nocs & 4) { c0 *= 0.5f; c1 *= 0.5f; c2 *= 0.5f; }
                gx += ((output_nocs & 3) == 2) ? c0 : -c0; gy += c1; gz += c2;
            }
            if (g_points) { g_points[e] = gx; g_points[e + 1] = gy; g_points[e + 2] = gz; }
            acc[0] += ax * x + bx * nx; acc[1] += ax * y + bx * ny; acc[2] += ax * z + bx * nz; acc[3] += ax;
            acc[4] += ay * x + by * nx; acc[5] += ay * y + by * ny; acc[6] += ay * z + by * nz; acc[7] += ay;
            acc[8] += az * x + bz * nx; acc[9] += az * y + bz * ny; acc[10] += az * z + bz * nz; acc[11] += az;
            if (LAT) {
                const float gs = -(gx * nx + gy * ny + gz * nz);               // grid.py:61 backward: d p / d sdf = -n_hat
#pragma unroll
                for (int i = 0; i < LAT_MAXL; ++i)
                    if (i < L) lat[i] += gs * jl[u][i];
            }
        }
    }
    __shared__ float red[12][PLB_THREADS / 64];
    __shared__ float redl[PLB_THREADS / 64];
    __shared__ float s_latn[LAT_MAXL];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) red[i][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 12) {
        float v = 0.f;
        for (int w = 0; w < PLB_THREADS / 64; ++w) v += red[tid][w];
        g_pose[(int64_t)b * 16 + tid] = v;
        red[tid][0] = v;                                               // the pose sums stay here for the parameter gradients below
    } else if (tid < 16) {
        g_pose[(int64_t)b * 16 + tid] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < LAT_MAXL; ++i) {
        if (i >= L || !LAT) break;
        float v = lat[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) redl[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < PLB_THREADS / 64; ++w) t += redl[w];
            g_latn[b * L + i] = t;
            s_latn[i] = t;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float g0 = red[0][0], g2 = red[2][0], g8 = red[8][0], g10 = red[10][0];
        const float c = cosf(yaw[b]), s = sinf(yaw[b]);
        g_yaw[b] = (-s) * g0 + c * g2 + (-c) * g8 + (-s) * g10;
        g_trans[b * 3] = red[3][0]; g_trans[b * 3 + 1] = red[7][0]; g_trans[b * 3 + 2] = red[11][0];
        if (LAT) {
            const float nrm = latnorm[b];
            float dot = 0.f;
            for (int i = 0; i < L; ++i) dot += (latent[b * L + i] / nrm) * s_latn[i];
            for (int i = 0; i < L; ++i) g_latent[b * L + i] = (s_latn[i] - (latent[b * L + i] / nrm) * dot) / nrm;
        } else {                                                        // pose-only: the latent is not a variable
            for (int i = 0; i < L; ++i) { g_latn[b * L + i] = 0.f; g_latent[b * L + i] = 0.f; }
        }
        if (SOLVE)           // the crop's solver step, reading the gradients this thread has just stored (and g_scale from the loss launch)
            sdfr_solver_crop(b, SA.B, SA.params, SA.grads, L, SA.loss2d, SA.loss3d, SA.npairs, SA.w2, SA.w3, SA.adam_m, SA.adam_v, SA.adam_t,
                             SA.lr_adam, SA.lr_scale, SA.lr_latent, SA.total, SA.stepped);
    }
}

extern "C" int sdfr_pose_latent_backward(const float* pose, const float* points, const float* normals, const float* g_p_cam,
                                         const float* g_n_cam, const float* g_col, int B, int cap, const int32_t* cnt, int output_nocs,
                                         const float* g_xyzf, const int32_t* fslot, const float* J, int n_inputs, int L, const float* yaw,
                                         const float* latent, const float* latnorm, float* g_points, float* g_pose, float* g_latn,
                                         float* g_yaw, float* g_trans, float* g_latent, void* stream) {
    // J == NULL: pose gradients only (pose-only refinement); g_latn / g_latent are zero-filled
    SDFR_REQUIRE(pose && points && normals && yaw && latent && latnorm && g_pose && g_latn && g_yaw && g_trans && g_latent,
                 "sdfr_pose_latent_backward: NULL argument");
    SDFR_REQUIRE(L >= 1 && L <= LAT_MAXL && L <= n_inputs, "sdfr_pose_latent_backward: latent size %d outside [1,%d] (use the separate kernels)", L,
                 LAT_MAXL);
    SDFR_REQUIRE(!g_xyzf || fslot, "sdfr_pose_latent_backward: g_xyzf needs fslot");
    SDFR_REQUIRE(output_nocs != 0, "sdfr_pose_latent_backward: built for the NOCS colour modes of the refinement loop");
    if (B <= 0) return SDFR_OK;
#define PLB_LAUNCH(X, LT)                                                                                                               \
    hipLaunchKernelGGL((sdfr_pose_latent_backward_kernel<X, LT>), dim3(B), dim3(PLB_THREADS), 0, (hipStream_t)stream, pose, points, normals, \
                       g_p_cam, g_n_cam, g_col, cap, cnt, output_nocs, g_xyzf, fslot, J, n_inputs, L, yaw, latent, latnorm, g_points, g_pose, \
                       g_latn, g_yaw, g_trans, g_latent)
    if (g_xyzf) { if (J) PLB_LAUNCH(true, true); else PLB_LAUNCH(true, false); }
    else { if (J) PLB_LAUNCH(false, true); else PLB_LAUNCH(false, false); }
#undef PLB_LAUNCH
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// sdfr_pose_latent_backward + the un-normalised xyzf gradient's factor + sdfr_solver_step in one launch (r06).  g_yaw / g_trans / g_latent must
// be the sections of `grads` (the flat [ yaw(B) | trans(B,3) | scale(B) | latent(B,L) ] buffer), g_scale its scale section as the loss wrote it.
extern "C" int sdfr_pose_latent_solver(const float* pose, const float* points, const float* normals, const float* g_p_cam, const float* g_n_cam,
                                       const float* g_col, int B, int cap, const int32_t* cnt, int output_nocs, const float* g_xyzf,
                                       const int32_t* fslot, const float* kscale, const float* J, int n_inputs, int L, const float* yaw,
                                       const float* latent, const float* latnorm, float* g_pose, float* g_latn, float* params, float* grads,
                                       const float* loss2d, const float* loss3d, const int32_t* npairs, float w2, float w3, float* adam_m,
                                       float* adam_v, int32_t* adam_t, float lr_adam, float lr_scale, float lr_latent, float* total,
                                       int32_t* stepped, void* stream) {
    SDFR_REQUIRE(pose && points && normals && yaw && latent && latnorm && g_pose && g_latn && g_xyzf && fslot && kscale && params && grads &&
                 loss2d && loss3d && npairs && adam_m && adam_v && adam_t && total && stepped, "sdfr_pose_latent_solver: NULL argument");
    SDFR_REQUIRE(L >= 1 && L <= LAT_MAXL && L <= n_inputs, "sdfr_pose_latent_solver: latent size %d outside [1,%d]", L, LAT_MAXL);
    SDFR_REQUIRE(output_nocs != 0, "sdfr_pose_latent_solver: built for the NOCS colour modes of the refinement loop");
    if (B <= 0) return SDFR_OK;
    SolveArgs SA = {kscale, params, grads, loss2d, loss3d, npairs, w2, w3, adam_m, adam_v, adam_t, lr_adam, lr_scale, lr_latent, B, total, stepped};
    float* g_yaw = grads; float* g_trans = grads + B; float* g_latent = grads + (int64_t)5 * B;
    if (J)
        hipLaunchKernelGGL((sdfr_pose_latent_backward_kernel<true, true, true>), dim3(B), dim3(PLB_THREADS), 0, (hipStream_t)stream, pose, points, normals,
                           g_p_cam, g_n_cam, g_col, cap, cnt, output_nocs, g_xyzf, fslot, J, n_inputs, L, yaw, latent, latnorm, (float*)nullptr,
                           g_pose, g_latn, g_yaw, g_trans, g_latent, SA);
    else
        hipLaunchKernelGGL((sdfr_pose_latent_backward_kernel<true, false, true>), dim3(B), dim3(PLB_THREADS), 0, (hipStream_t)stream, pose, points, normals,
                           g_p_cam, g_n_cam, g_col, cap, cnt, output_nocs, g_xyzf, fslot, J, n_inputs, L, yaw, latent, latnorm, (float*)nullptr,
                           g_pose, g_latn, g_yaw, g_trans, g_latent, SA);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}


// ---- the decoder's scale head (deep_sdf_decoder_scale.py:68-75,110-112): Linear(L,3) - ReLU - Linear(3,3) - ReLU - Linear(3,1) on ONE latent row.
// The reference evaluates it in every Decoder.forward and returns it next to the SDF values (the refinement loop ignores it,
// pipelines/optimizer.py:101); as five ATen launches it cost more host time per iteration than the band kernels.  One thread.
__global__ void sdfr_scale_net_kernel(const float* __restrict__ lat, int L, const float* __restrict__ W1, const float* __restrict__ b1,
                                      const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3,
                                      const float* __restrict__ b3, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float h1[3], h2[3];
    for (int j = 0; j < 3; ++j) {
        float s = 0.f;
        for (int k = 0; k < L; ++k) s += W1[j * L + k] * lat[k];
        h1[j] = fmaxf(s + b1[j], 0.f);
    }
    for (int j = 0; j < 3; ++j) {
        float s = 0.f;
        for (int k = 0; k < 3; ++k) s += W2[j * 3 + k] * h1[k];
        h2[j] = fmaxf(s + b2[j], 0.f);
    }
    float s = 0.f;
    for (int k = 0; k < 3; ++k) s += W3[k] * h2[k];
    out[0] = s + b3[0];
}

extern "C" int sdfr_scale_net(const float* latent_row, int L, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                              const float* b3, float* out, void* stream) {
    SDFR_REQUIRE(latent_row && W1 && b1 && W2 && b2 && W3 && b3 && out && L > 0, "sdfr_scale_net: bad argument");
    hipLaunchKernelGGL(sdfr_scale_net_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, latent_row, L, W1, b1, W2, b2, W3, b3, out);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
