// Conservative screen box of a disc surfel (shared by the splat kernels and the fused surfel-forward kernel of project.hip).
#pragma once
#include "sdfr_common.h"

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
