// Decoder forward on the grid with ERROR-COMPENSATED float16 operands ("split" forward), padded hidden width 512; compiled alone.
//
// gfx950 has no reduced-precision f32 matrix mode: the exact v_mfma_f32_32x32x2_f32 path tops out at 157 TFLOP/s while the f16 matrix
// cores deliver 16x that.  Here every float32 operand x (weights, packed once on the host; activations, split in the layer epilogue)
// is carried as two halves
//        hi = half(x)                     11 significand bits
//        lo = half((x - hi) * 2^11)       the next 11 bits, pre-scaled so that it never falls into the half subnormal range
// and a product  a*b  is evaluated as   a_hi*b_hi  +  2^-11 * (a_hi*b_lo + a_lo*b_hi)   -- three v_mfma_f32_32x32x16_f16 with float32
// accumulation into two accumulator sets (main, correction) instead of 16 passes of the f32 instruction.  The dropped term a_lo*b_lo
// is below 2^-22 relative, i.e. the result carries ~22 significand bits per product against float32's 24: the deviation from
// sdfr_mlp_forward is of the order of float32 summation-order noise (measured in tests/test_gpu_parity.py), not of half precision.
// Requires |activation| < 65504 (half range of `hi`).
//
// Measured (MI355X, 64000 rows, 8x512 decoder): 0.71 ms against 1.82 ms of the exact-f32 kernel; max |error| against a float64
// evaluation 1.6e-7 (exact-f32 kernel: 1.5e-7; plain f16 kernel: 3.9e-4).  The matrix pipe is ~50 % busy: like the plain f16 kernel this
// one is paced by the weight stream through the CU's 64 B/clk vector-memory path (1 MiB of hi+lo weights per layer per 64 points needs
// 2/3 of that path at full MFMA rate); a larger point tile would need more than the CU's 160 KiB of LDS for the operand planes.
//
// Geometry is the f32 forward's (8 waves, 64-point tiles, wave w owns features [64w, 64w+64)); the ReLU masks are saved in the library's
// one layout (mlp_kernel.h: sdfr_mask_dword) and the mask-fed Jacobian kernels consume them like any other forward's.  LDS: 128 KiB of operands (hi and lo rows interleaved).
// Follows Decoder.forward, reference sdfrenderer/deepsdf/networks/deep_sdf_decoder_scale.py:78-107.
#include "mlp_kernel.h"
#ifndef SDFR_S_PF
#define SDFR_S_PF 2
#endif
#ifndef SDFR_S_PFB
#define SDFR_S_PFB 2
#endif

template <int FT, int NP, int NW, int PF, int PFB, bool SAVE>
__global__ __launch_bounds__(64 * NW, 1) void sdfr_mlp_split_kernel(const MlpParams P) {
    constexpr int MS = 32, KV = 8, NLG = 2, RG = 4, KT = KV * NLG;
    constexpr int NT = 64 * NW, PT = MS * NP, HP = MS * FT * NW, KG = HP / KV;
    constexpr int MW = FT * NP * RG * 4 / 32;                      // mask words per thread per layer (f32 forward layout)
    constexpr float UP = 2048.f, DOWN = 1.f / 2048.f;
    static_assert(PF % PFB == 0 && 4 % PF == 0, "PF must be a multiple of PFB and divide the 4-tile K padding");
    __shared__ float4 lds4[2 * KG * PT + NT / 4];
    // operand planes interleaved per k group: act[k/8][hi|lo][point] 16-byte vectors (the lo row sits PT vectors = 1 KiB after the hi row,
    // inside the ds_read immediate-offset range; each row is contiguous over points: conflict-free ds_read_b128)
    h16x8* act = reinterpret_cast<h16x8*>(lds4);
    h16* act_e = reinterpret_cast<h16*>(lds4);
    float* red = reinterpret_cast<float*>(lds4 + 2 * KG * PT);    // [NT]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lp = lane % MS, lg = lane / MS;
    const int NI = P.n_inputs;
    const int64_t r0 = (int64_t)blockIdx.x * PT;
    const int64_t n_rows = P.n_dev ? min(P.n, (int64_t)*P.n_dev) : P.n;       // (optional device-side row count: sdfr_mlp_forward_split_counted)
    if (r0 >= n_rows) return;
    const int n_valid = (int)min((int64_t)PT, n_rows - r0);
    auto row_of = [&](int pt) { return r0 + (pt < n_valid ? pt : 0); };
    auto elem = [&](int k, int pt) { return (((k / KV) * 2) * PT + pt) * KV + (k % KV); };      // hi element; lo is PT * KV further

    // ---- layer-0 operand ----------------------------------------------------------------------------------
    {
        const int k0pad = P.L[0].kp_s;
        for (int e = tid; e < PT * k0pad; e += NT) {
            const int pt = e / k0pad, k = e - pt * k0pad;
            const float v = (k < NI) ? P.inputs[row_of(pt) * NI + k] : 0.f;
            const h16 h = (h16)v;
            act_e[elem(k, pt)] = h;
            act_e[elem(k, pt) + PT * KV] = (h16)((v - (float)h) * UP);
        }
    }
    __syncthreads();

    const int fbase = wave * MS * FT;
    f32x16 acc[FT][NP], cor[FT][NP];

    // One transposed product over `nkt` K tiles (a multiple of PF; the host pads every layer's K to 64).  Straight-line body: the
    // prefetch index is clamped instead of guarded, so the loop has no branches and the compiler's wait counts stay exact.
    auto gemm = [&](const h16x8* __restrict__ Wl, int nkt) {
        const h16x8* ap = Wl + (lg * HP + fbase + lp) * 2;        // [k/8][row][hi|lo]: a lane's hi and lo fragments are adjacent (32 bytes)
        const h16x8* bp = act + lg * 2 * PT + lp;
        h16x8 ah[PF][FT], al[PF][FT], bh[PFB][NP], bl[PFB][NP];
        auto load_a = [&](int tile, h16x8* hh, h16x8* ll) {
            const h16x8* o = ap + (int64_t)tile * (NLG * HP * 2);
#pragma unroll
            for (int f = 0; f < FT; ++f) { hh[f] = o[f * MS * 2]; ll[f] = o[f * MS * 2 + 1]; }
        };
        auto load_b = [&](int tile, h16x8* hh, h16x8* ll) {
            const h16x8* o = bp + tile * (NLG * 2 * PT);
#pragma unroll
            for (int p = 0; p < NP; ++p) { hh[p] = o[p * MS]; ll[p] = o[PT + p * MS]; }
        };
        const int last = nkt - 1;
#pragma unroll
        for (int u = 0; u < PF - 1; ++u) load_a(min(u, last), ah[u], al[u]);
#pragma unroll
        for (int u = 0; u < PFB - 1; ++u) load_b(min(u, last), bh[u], bl[u]);
        for (int t = 0; t < nkt; t += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                load_a(min(t + u + PF - 1, last), ah[(u + PF - 1) % PF], al[(u + PF - 1) % PF]);
                load_b(min(t + u + PFB - 1, last), bh[(u + PFB - 1) % PFB], bl[(u + PFB - 1) % PFB]);
                // keep the prefetches ahead of this tile's products: left free, the scheduler sinks each load next to its first use to
                // save registers and every product then waits a full L2 round trip
                __builtin_amdgcn_sched_barrier(0);
                // three products per accumulator pair, issued set by set so that no instruction waits on its predecessor
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        acc[f][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][f], bh[u % PFB][p], acc[f][p], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        cor[f][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][f], bl[u % PFB][p], cor[f][p], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        cor[f][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u][f], bh[u % PFB][p], cor[f][p], 0, 0, 0);
            }
        }
    };
    auto feat0 = [&](int f, int rg) { return fbase + f * MS + rg * (4 * NLG) + 4 * lg; };
    const h16x8* Ws = reinterpret_cast<const h16x8*>(P.Ws);

    for (int l = 0; l < P.n_mfma; ++l) {
        const MlpLayer L = P.L[l];
        const MlpLayer Ln = P.L[l + 1];
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[f][p][r] = 0.f; cor[f][p][r] = 0.f; }
        // waves whose 64 feature rows lie beyond the layer's width skip the product (padded rows of a partly used block are zero)
        if (__builtin_amdgcn_readfirstlane(L.out_dim - fbase) > 0) gemm(Ws + L.off_s, L.kp_s / KT);
        const float* bias = P.bias + l * HP;
        float4 b4s[FT][RG];
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) b4s[f][rg] = *reinterpret_cast<const float4*>(bias + feat0(f, rg));
        __syncthreads();                                  // every wave is done reading the operand planes
        uint32_t mw[MW];
#pragma unroll
        for (int w = 0; w < MW; ++w) mw[w] = 0u;
        const int inj_lo = L.out_dim, inj_hi = L.out_dim + Ln.inj_n;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int j0 = feat0(f, rg);
                const float4 b4 = b4s[f][rg];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int pt = p * MS + lp;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = (acc[f][p][rg * 4 + i] + cor[f][p][rg * 4 + i] * DOWN) + f4c(b4, i);
                        const bool pos = x > 0.f;
                        v[i] = pos ? x : 0.f;
                        if (SAVE) {
                            const int bit = ((f * NP + p) * RG + rg) * 4 + i;
                            mw[bit >> 5] |= (pos ? 1u : 0u) << (bit & 31);
                        }
                    }
                    if (j0 + 3 >= inj_lo && j0 < inj_hi) {    // re-inject input columns for the next layer
                        const float* src = P.inputs + row_of(pt) * NI + Ln.inj_off - inj_lo;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (j0 + i >= inj_lo && j0 + i < inj_hi) v[i] = src[j0 + i];
                    }
                    h16x4 h4, l4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h4[i] = (h16)v[i];
                        l4[i] = (h16)((v[i] - (float)h4[i]) * UP);
                    }
                    const int e = elem(j0, pt);
                    *reinterpret_cast<h16x4*>(act_e + e) = h4;
                    *reinterpret_cast<h16x4*>(act_e + e + PT * KV) = l4;
                }
            }
        if (SAVE && P.maskbuf) {
            // mask layout v2 (mlp_kernel.h: sdfr_mask_dword): this thread's 16 bits of (feature tile f, point tile p) are one short of row
            // r0 + p*32 + lp, dword (fbase >> 5) + f, half lg
            if constexpr (FT == 2) {
                // (as in mlp_kernel.h: lanes lp and lp + 32 exchange their halves once per point tile; one of them stores the wave's 8 bytes of the row)
                uint2* mb2 = reinterpret_cast<uint2*>(P.maskbuf) + (((r0 >> 7) * P.n_mfma + l) * 128) * (int64_t)(HP / 64);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const uint32_t h0 = (mw[p >> 1] >> ((p & 1) * 16)) & 0xffffu;
                    const uint32_t h1 = (mw[(NP + p) >> 1] >> (((NP + p) & 1) * 16)) & 0xffffu;
                    const uint32_t mine = h0 | (h1 << 16);
                    const uint32_t other = (uint32_t)__shfl_xor((int)mine, 32, 64);
                    const uint32_t lo = lg == 0 ? mine : other, hi = lg == 0 ? other : mine;
                    if (lg == (p & 1))
                        mb2[((int)(r0 & 127) + p * MS + lp) * (HP / 64) + (fbase >> 6)] = make_uint2((lo & 0xffffu) | (hi << 16), (lo >> 16) | (hi & 0xffff0000u));
                }
            } else {
            uint16_t* mb = reinterpret_cast<uint16_t*>(P.maskbuf) + (((r0 >> 7) * P.n_mfma + l) * 128) * (int64_t)(HP / 32) * 2;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int off = (((int)(r0 & 127) + p * MS + lp) * (HP / 32) + (fbase >> 5)) * 2 + lg;
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const int fi = f * NP + p;
                    mb[off + 2 * f] = (uint16_t)(mw[fi >> 1] >> ((fi & 1) * 16));
                }
            }
            }
        }
        __syncthreads();
    }

    // ---- last linear (H -> 1) + tanh, float32 on the recombined operand ------------------------------------------------
    {
        constexpr int SL = NT / PT, KGS = KG / SL;
        const int sl = tid / PT, pt = tid - sl * PT;
        const float* wl = P.w_last + sl * KGS * KV;
        float s = 0.f;
#pragma unroll 4
        for (int g = 0; g < KGS; ++g) {
            const h16x8 a = act[(sl * KGS + g) * 2 * PT + pt];
            const h16x8 b = act[(sl * KGS + g) * 2 * PT + PT + pt];
#pragma unroll
            for (int i = 0; i < KV; ++i) s = fmaf((float)a[i] + (float)b[i] * DOWN, wl[g * KV + i], s);
        }
        red[tid] = s;
        __syncthreads();
        if (tid < PT) {
            float y = 0.f;
#pragma unroll
            for (int q = 0; q < SL; ++q) y += red[q * PT + tid];
            y += P.b_last;
            const float y1 = P.use_tanh ? tanhf(y) : y;
            if (tid < n_valid) P.sdf[r0 + tid] = tanhf(y1);
        }
    }
}

void sdfr_launch_fwd_split_512(const MlpParams& P, int64_t n, bool save_masks, hipStream_t s) {
    const int grid = sdfr_cdiv(n, 64);
    if (save_masks)
        hipLaunchKernelGGL((sdfr_mlp_split_kernel<2, 2, 8, SDFR_S_PF, SDFR_S_PFB, true>), dim3(grid), dim3(512), 0, s, P);
    else
        hipLaunchKernelGGL((sdfr_mlp_split_kernel<2, 2, 8, SDFR_S_PF, SDFR_S_PFB, false>), dim3(grid), dim3(512), 0, s, P);
}
