// The MultipleOptimizer step of the reference's refinement loop for ONE crop (pipelines/optimizer.py:13-23,44-52): Adam(lr .01) on yaw and
// trans, SGD(lr .01 / 3e-5) on scale / latent, gated by the loop's skip conditions (:127-129,149-151).  Shared by sdfr_solver_step
// (losses.hip) and the fused backward tail sdfr_pose_latent_solver (params.hip, r06): the same statements, the same bits.
// params / grads: one flat structure-of-arrays buffer  [ yaw(B) | trans(B,3) | scale(B) | latent(B,L) ].
#pragma once
#include "sdfr_common.h"

__device__ __forceinline__ void sdfr_solver_crop(const int b, const int B, float* params, const float* grads, int L,   /* (no __restrict__: the fused tail has just written grads through other pointers) */
                                                 const float* __restrict__ loss2d, const float* __restrict__ loss3d,
                                                 const int32_t* __restrict__ npairs, float w2, float w3, float* __restrict__ adam_m,
                                                 float* __restrict__ adam_v, int32_t* __restrict__ adam_t, float lr_adam, float lr_scale,
                                                 float lr_latent, float* __restrict__ total, int32_t* __restrict__ stepped) {
    const float l = w3 * loss3d[b] + w2 * loss2d[b];                    // :144-146
    total[b] = l;
    const bool skip = (npairs[b] < 0) || isnan(l) || (l == 0.f);        // :127-129, :149-151
    stepped[b] = skip ? 0 : 1;
    if (skip) return;
    // section offsets of the structure-of-arrays buffer
    auto at = [&](int i) -> int64_t {              // i: 0 yaw, 1..3 trans, 4 scale, 5.. latent
        if (i == 0) return b;
        if (i < 4) return (int64_t)B + (int64_t)b * 3 + (i - 1);
        if (i == 4) return (int64_t)4 * B + b;
        return (int64_t)5 * B + (int64_t)b * L + (i - 5);
    };
    float* p = params;
    const float* g = grads;
    const int t = adam_t[b] + 1;
    adam_t[b] = t;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(b2, (float)t);
    for (int i = 0; i < 4; ++i) {                                        // Adam on yaw, trans (:34-36,47-49)
        float m = adam_m[b * 4 + i], v = adam_v[b * 4 + i];
        const float gi = g[at(i)];
        m = b1 * m + (1.f - b1) * gi;
        v = b2 * v + (1.f - b2) * gi * gi;
        adam_m[b * 4 + i] = m; adam_v[b * 4 + i] = v;
        const float denom = sqrtf(v) / sqrtf(bc2) + eps;
        p[at(i)] -= (lr_adam / bc1) * (m / denom);
    }
    p[at(4)] -= lr_scale * g[at(4)];                                      // SGD on scale, latent (:37-38,50-51)
    for (int i = 0; i < L; ++i) p[at(5 + i)] -= lr_latent * g[at(5 + i)];
}
