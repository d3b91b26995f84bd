// DeepSDF decoder on MI355X (gfx950): fused multi-layer MLP forward and input-Jacobian backward.
//
// Replaces Decoder.forward (reference sdfrenderer/deepsdf/networks/deep_sdf_decoder_scale.py:78-107) and the
// autograd backward of it w.r.t. its input rows (the normals hook of sdfrenderer/grid.py:55-56 and the latent
// gradient of pipelines/optimizer.py:156).
//
// Design (CDNA4-first, nothing here is translated from the reference's ATen graph):
//   * One 256-thread workgroup (4 waves, one per SIMD) owns a tile of PT = 32*NP grid points and carries them
//     through EVERY layer.  Activations never leave the CU: they live in LDS as  act[k/4][point][k%4]  (float4),
//     128 KiB for 512 features x 64 points.
//   * Each layer is computed transposed,  out^T[feature][point] = W[feature][k] * act^T[k][point],  with the exact-f32
//     matrix instruction v_mfma_f32_32x32x2_f32.  Wave w owns output features [w*32*FT, (w+1)*32*FT): FT x NP
//     32x32 accumulator tiles (128 VGPRs at FT=4, NP=2).  The MFMA A operand (weights) is NOT shared between waves,
//     so it is streamed straight from L2 into VGPRs (no LDS staging, no barrier in the K loop) from a tile-major
//     image packed once at load time:  Wf[tile t][kg 0..1][row 0..HP)[4] = W[row][8t + 4kg + 0..3]  -- each lane's
//     fragment is one coalesced 16-byte load, and one load feeds four MFMA k-steps.  The B operand (activations) is
//     one conflict-free ds_read_b128 per 32 points.  With the transposed product a lane's 4 consecutive accumulator
//     registers are 4 consecutive features of ONE point, so the epilogue (bias + ReLU + latent re-injection) writes
//     the next layer's operand with conflict-free ds_write_b128.  Only two barriers per layer.
//   * The last linear (H -> 1) is a VALU dot product out of LDS followed by tanh.
//   * Jacobian mode (JAC): the same kernel keeps the ReLU masks (1 bit per feature per point) in LDS and runs the
//     layers backwards with the transposed weight image Wb, giving d sdf / d input-row for the selected rows only.
//     The backward needs no activations, only masks, because just the INPUT gradient is required (weights are frozen).
#include "sdfr_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));

struct MlpLayer {
    int in_dim, out_dim;    // true widths (in_dim includes injected input columns)
    int inj_n, inj_off;     // input columns concatenated BEFORE this layer
    int kp_f, kp_b, kp_h;   // padded K extents: forward f32 (in_dim -> x16), backward f32 (out_dim -> x16), forward f16 (in_dim -> x32)
    int off_f, off_b, off_h;// 16-byte-vector offsets of the layer's image in Wf / Wb / Wh
};

struct MlpParams {
    const float4* Wf;       // forward image, float32:  [k/4][HP] float4
    const float4* Wb;       // backward (transposed) image, float32
    const void* Wh;         // forward image, float16:  [k/8][HP] 8 x half
    int fwd_np;             // MODE 3: point tiles per workgroup of the forward launch that saved the masks (2: f32, 4: f16)
    const float* bias;      // [n_mfma][HP]
    const float* w_last;    // [HP] zero padded
    float b_last;
    int n_mfma;             // layers computed with MFMA = n_lin - 1
    int n_inputs;
    int use_tanh;
    MlpLayer L[SDFR_MAX_LAYERS];
    // forward mode
    const float* inputs;
    int64_t n;
    float* sdf;
    // jacobian mode
    int64_t rows_per_crop;
    const int32_t* idx;
    const int32_t* cnt;
    int cap;
    float* J;
    float* sdf_sel;
    const float* sdf_in;    // MODE 3: decoder output of the forward launch that saved the masks
    uint32_t* maskbuf;      // MODE 1 (write) / MODE 3 (read): ReLU masks [tile64][layer][word][thread]
};

struct sdfr_decoder {
    int device;
    int n_lin, n_inputs, use_tanh, HP;
    float4* d_Wf;
    float4* d_Wb;
    void* d_Wh;
    float* d_bias;
    float* d_wlast;
    int64_t macs;
    MlpParams proto;
};

__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// Matrix instruction + operand vector per element type ET and output tile MS x MS.  A lane's operand fragment is one 16-byte vector:
//   float : 4 consecutive k; exact-f32 v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, 4 instructions per fragment (one per component)
//   half  : 8 consecutive k; v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 (f32 accumulate), 1 instruction per fragment
template <typename ET, int MS> struct Mma;
template <> struct Mma<float, 32> {
    typedef f32x16 acc_t; typedef float4 vec_t;
    static constexpr int KV = 4, NSTEP = 4;
    static __device__ __forceinline__ acc_t step(const vec_t& a, const vec_t& b, acc_t c, int ks) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a, ks), f4c(b, ks), c, 0, 0, 0);
    }
};
template <> struct Mma<float, 16> {
    typedef f32x4 acc_t; typedef float4 vec_t;
    static constexpr int KV = 4, NSTEP = 4;
    static __device__ __forceinline__ acc_t step(const vec_t& a, const vec_t& b, acc_t c, int ks) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a, ks), f4c(b, ks), c, 0, 0, 0);
    }
};
template <> struct Mma<h16, 32> {
    typedef f32x16 acc_t; typedef h16x8 vec_t;
    static constexpr int KV = 8, NSTEP = 1;
    static __device__ __forceinline__ acc_t step(const vec_t& a, const vec_t& b, acc_t c, int) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<h16, 16> {
    typedef f32x4 acc_t; typedef h16x8 vec_t;
    static constexpr int KV = 8, NSTEP = 1;
    static __device__ __forceinline__ acc_t step(const vec_t& a, const vec_t& b, acc_t c, int) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
__device__ __forceinline__ void store4(float* dst, const float* v) { *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(h16* dst, const float* v) {
    h16x4 t; t[0] = (h16)v[0]; t[1] = (h16)v[1]; t[2] = (h16)v[2]; t[3] = (h16)v[3];
    *reinterpret_cast<h16x4*>(dst) = t;
}

// ET   operand element type (float: exact f32; _Float16: half operands, f32 accumulate -- forward modes only)
// MS   MFMA tile (32: forward on the grid; 16: small tiles so that a few thousand band rows fill the chip)
// FT   feature tiles (MS rows) per wave, NP point tiles (MS points) per workgroup, NW waves per workgroup (HP = MS*FT*NW padded width)
// PF   weight/activation fragment buffers in flight per wave (prefetch distance PF-1 K tiles)
// MODE 0: forward.  1: forward + ReLU masks saved to HBM (1 bit per feature, point, layer).  2: Jacobian of selected rows by
//      recomputation (forward with masks in LDS, then backward).  3: Jacobian of selected rows from the masks a MODE-1 launch saved
//      (backward only: no activations are needed for an input gradient, only the masks and the output).
// Lane map of one MS x MS accumulator tile: point = lane % MS, feature = (reg/4)*4*NLG + 4*(lane/MS) + reg%4 with NLG = 64/MS lane
// groups; a lane's 4 consecutive registers are 4 consecutive features of one point.
template <typename ET, int MS, int FT, int NP, int NW, int PF, int MODE>
__global__ __launch_bounds__(64 * NW) void sdfr_mlp_kernel(const MlpParams P) {
    typedef Mma<ET, MS> M;
    typedef typename M::acc_t acc_t;
    typedef typename M::vec_t vec_t;
    constexpr bool HALF = sizeof(ET) == 2;
    constexpr bool JAC = MODE >= 2;
    constexpr bool SAVE = MODE == 1;
    constexpr bool LMASK = MODE == 2;
    constexpr bool GMASK = MODE == 3;
    static_assert(!SAVE || MS == 32, "mask layout assumes 32x32 forward tiles");
    static_assert(!HALF || !JAC, "the Jacobian modes are float32");
    constexpr int KV = M::KV;                                      // operand elements per 16-byte fragment
    constexpr int NLG = 64 / MS;                                   // lane groups (k slots per MFMA)
    constexpr int RG = MS / (4 * NLG);                             // register groups of 4 per accumulator (4 or 1)
    constexpr int KT = KV * NLG;                                   // k per fragment tile
    constexpr int NT = 64 * NW;
    constexpr int PT = MS * NP;
    constexpr int HP = MS * FT * NW;
    constexpr int KG = HP / KV;                                    // 16-byte k groups of the activation tile
    constexpr int FT32 = HP / (32 * NW);                           // feature tiles per wave of the 32x32 forward kernel (mask layout)
    constexpr int MW = (FT * NP * RG * 4 + 31) / 32;               // mask words per thread per layer
    constexpr int MASK_WORDS = LMASK ? (SDFR_MAX_LAYERS * MW * NT) : 1;
    // single LDS object, carved by hand (16-byte aligned pieces first)
    __shared__ float4 lds4[KG * PT + NT / 4 + 32 + (PT + 3) / 4 * 2 + (MASK_WORDS + 3) / 4];
    vec_t* act = reinterpret_cast<vec_t*>(lds4);                  // [KG][PT] 16-byte vectors: act[k/KV][point][k%KV]
    ET* act_e = reinterpret_cast<ET*>(lds4);
    float* red = reinterpret_cast<float*>(lds4 + KG * PT);        // [NT]
    int* rows = reinterpret_cast<int*>(lds4 + KG * PT + NT / 4);  // [PT] source row of each point (128 ints max)
    float* gy = reinterpret_cast<float*>(lds4 + KG * PT + NT / 4 + 32);   // [PT] d out / d y_last
    int* slots = reinterpret_cast<int*>(lds4 + KG * PT + NT / 4 + 32 + (PT + 3) / 4);   // [PT] J slot or -1
    uint32_t* masks = reinterpret_cast<uint32_t*>(lds4 + KG * PT + NT / 4 + 32 + (PT + 3) / 4 * 2);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lp = lane % MS;              // point within a point tile
    const int lg = lane / MS;              // lane group: k slot of the operands, feature sub-block of the accumulator
    const int NI = P.n_inputs;

    // ---- which rows does this tile hold ------------------------------------------------------------------
    int n_valid;
    if (JAC) {
        const int b = blockIdx.y;
        const int count = sdfr_count(P.cnt, b, P.cap);
        const int s0 = blockIdx.x * PT;
        if (s0 >= count) return;
        n_valid = min(PT, count - s0);
        if (tid < PT) {
            const bool v = tid < n_valid;
            const int s = v ? (s0 + tid) : s0;
            rows[tid] = (int)(P.rows_per_crop * b) + P.idx[(int64_t)b * P.cap + s];
            slots[tid] = v ? (b * P.cap + s) : -1;
        }
    } else {
        const int64_t r0 = (int64_t)blockIdx.x * PT;
        n_valid = (int)min((int64_t)PT, P.n - r0);
        if (tid < PT) rows[tid] = (int)(r0 + (tid < n_valid ? tid : 0));
    }
    __syncthreads();

    // ---- layer-0 operand: act[k][pt] = inputs[row(pt)][k], zero padded to the K tile ---------------------
    if (!GMASK) {
        const int k0pad = HALF ? P.L[0].kp_h : P.L[0].kp_f;
        for (int e = tid; e < PT * k0pad; e += NT) {
            const int pt = e / k0pad, k = e - pt * k0pad;
            const float v = (k < NI) ? P.inputs[(int64_t)rows[pt] * NI + k] : 0.f;
            act_e[((k / KV) * PT + pt) * KV + (k % KV)] = (ET)v;
        }
    }
    __syncthreads();

    const int fbase = wave * MS * FT;      // first feature row owned by this wave
    acc_t acc[FT][NP];

    // One transposed GEMM over a K extent of `kpad` (a multiple of KT): acc[f][p] += W_tile(rows fbase+f*MS..) x act.
    // FULL: all FT feature tiles of this wave are active (straight-line MFMA stream, no branches);
    // otherwise only the first `nact` tiles are (thin layers: the 6-wide first layer's backward, small nets).
    auto gemm_body = [&](const vec_t* __restrict__ Wl, int kpad, int nact, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nkt = kpad / KT;
        const vec_t* aptr = Wl + lg * HP + fbase + lp;
        const vec_t* bptr = act + lg * PT + lp;
        vec_t a[PF][FT], b[PF][NP];
        auto load = [&](int tile, vec_t* aa, vec_t* bb) {
            const vec_t* ap = aptr + (int64_t)tile * (NLG * HP);
            const vec_t* bp = bptr + tile * (NLG * PT);
#pragma unroll
            for (int f = 0; f < FT; ++f)
                if (FULL || f < nact) aa[f] = ap[f * MS];
#pragma unroll
            for (int p = 0; p < NP; ++p) bb[p] = bp[p * MS];
        };
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int i = 0; i < KV; ++i) a[u][f][i] = (ET)0;
#pragma unroll
        for (int u = 0; u < PF - 1; ++u)
            if (u < nkt) load(u, a[u], b[u]);
        for (int t = 0; t < nkt; t += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (t + u < nkt) {
                    if (t + u + PF - 1 < nkt) load(t + u + PF - 1, a[(u + PF - 1) % PF], b[(u + PF - 1) % PF]);
#pragma unroll
                    for (int ks = 0; ks < M::NSTEP; ++ks)
#pragma unroll
                        for (int f = 0; f < FT; ++f)
                            if (FULL || f < nact) {
#pragma unroll
                                for (int p = 0; p < NP; ++p) acc[f][p] = M::step(a[u][f], b[u][p], acc[f][p], ks);
                            }
                }
            }
        }
    };
    auto gemm = [&](const vec_t* __restrict__ Wl, int kpad, int rows_active) {
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < RG * 4; ++r) acc[f][p][r] = 0.f;
        int nact = (rows_active - fbase + MS - 1) / MS;
        nact = nact < 0 ? 0 : (nact > FT ? FT : nact);
        nact = __builtin_amdgcn_readfirstlane(nact);
        if (nact == FT) gemm_body(Wl, kpad, FT, std::true_type{});
        else if (nact > 0) gemm_body(Wl, kpad, nact, std::false_type{});
    };
    // first feature of the 4-register group rg of feature tile f held by this lane
    auto feat0 = [&](int f, int rg) { return fbase + f * MS + rg * (4 * NLG) + 4 * lg; };
    const vec_t* Wfwd = reinterpret_cast<const vec_t*>(HALF ? (const void*)P.Wh : (const void*)P.Wf);

    // ---- forward through the MFMA layers -------------------------------------------------------------------
    for (int l = 0; !GMASK && l < P.n_mfma; ++l) {
        const MlpLayer L = P.L[l];
        const MlpLayer Ln = P.L[l + 1];
        gemm(Wfwd + (HALF ? L.off_h : L.off_f), HALF ? L.kp_h : L.kp_f, L.out_dim);
        __syncthreads();                                  // every wave is done reading act
        uint32_t mw[MW];
#pragma unroll
        for (int w = 0; w < MW; ++w) mw[w] = 0u;
        const float* bias = P.bias + l * HP;
        const int inj_lo = L.out_dim, inj_hi = L.out_dim + Ln.inj_n;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int j0 = feat0(f, rg);
                const float4 b4 = *reinterpret_cast<const float4*>(bias + j0);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int pt = p * MS + lp;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = acc[f][p][rg * 4 + i] + f4c(b4, i);
                        const bool pos = x > 0.f;
                        v[i] = pos ? x : 0.f;
                        if (LMASK || SAVE) {
                            const int bit = ((f * NP + p) * RG + rg) * 4 + i;
                            mw[bit >> 5] |= (pos ? 1u : 0u) << (bit & 31);
                        }
                    }
                    if (j0 + 3 >= inj_lo && j0 < inj_hi) {    // re-inject input columns for the next layer
                        const float* src = P.inputs + (int64_t)rows[pt] * NI + Ln.inj_off - inj_lo;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (j0 + i >= inj_lo && j0 + i < inj_hi) v[i] = src[j0 + i];
                    }
                    store4(act_e + ((j0 / KV) * PT + pt) * KV + (j0 % KV), v);
                }
            }
        if (LMASK) {
#pragma unroll
            for (int w = 0; w < MW; ++w) masks[(l * MW + w) * NT + tid] = mw[w];
        }
        if (SAVE && P.maskbuf) {
            // [point tile][layer][word][thread]: bit ((f*NP + p)*4 + rg)*4 + i of the thread's mask (32x32 geometry)
            uint32_t* dst = P.maskbuf + (((int64_t)blockIdx.x * P.n_mfma + l) * MW) * NT + tid;
#pragma unroll
            for (int w = 0; w < MW; ++w) dst[w * NT] = mw[w];
        }
        __syncthreads();
    }

    // ---- last linear (H -> 1) + tanh -----------------------------------------------------------------------
    if (GMASK) {
        if (tid < PT) {
            const float o = P.sdf_in[rows[tid]];
            gy[tid] = 1.f - o * o;
            if (tid < n_valid && P.sdf_sel) P.sdf_sel[slots[tid]] = o;
        }
    } else {
        constexpr int SL = NT / PT;                       // k slices
        constexpr int KGS = KG / SL;
        const int sl = tid / PT, pt = tid - sl * PT;
        const float* wl = P.w_last + sl * KGS * KV;
        const vec_t* a4 = act + (sl * KGS) * PT + pt;
        float s = 0.f;
#pragma unroll 4
        for (int g = 0; g < KGS; ++g) {
            const vec_t a = a4[g * PT];
#pragma unroll
            for (int i = 0; i < KV; ++i) s = fmaf((float)a[i], wl[g * KV + i], s);
        }
        red[tid] = s;
        __syncthreads();
        if (tid < PT) {
            float y = 0.f;
#pragma unroll
            for (int q = 0; q < SL; ++q) y += red[q * PT + tid];
            y += P.b_last;
            const float y1 = P.use_tanh ? tanhf(y) : y;
            const float o = tanhf(y1);
            if (JAC) {
                float g = 1.f - o * o;
                if (P.use_tanh) g *= (1.f - y1 * y1);
                gy[tid] = g;
                if (tid < n_valid && P.sdf_sel) P.sdf_sel[slots[tid]] = o;
            } else {
                if (tid < n_valid) P.sdf[(int64_t)blockIdx.x * PT + tid] = o;
            }
        }
    }
    if constexpr (JAC) {
    __syncthreads();

    // ---- backward: d out / d inputs for every point of the tile ---------------------------------------------
    // in-gradient of layer l (features k = in-features of layer l) -> masked operand for layer l-1, or J
    auto store_in_grad = [&](int l, auto&& value) {
        const MlpLayer L = P.L[l];
        const int prev_out = P.L[l - 1].out_dim;
        const int inj_hi = prev_out + L.inj_n;
        uint32_t mw[MW];
        if (GMASK) {
            // decode the forward launch's layout (32x32 tiles, P.fwd_np point tiles per workgroup): feature jr of this wave, point q of
            // the forward tile -> bit ((jr/32 * np + q/32)*4 + (jr%32)/8)*4 + jr%4 of thread wave*64 + ((jr%32)/4 % 2)*32 + q%32
#pragma unroll
            for (int w = 0; w < MW; ++w) mw[w] = 0u;
            const int np = P.fwd_np;
            const int r = rows[lp];
            const int tile = r / (32 * np), q = r - tile * (32 * np);
            const int mwf = FT32 * np / 2;                                   // mask words per thread of the forward kernel
            const uint32_t* src = P.maskbuf + (((int64_t)tile * P.n_mfma + (l - 1)) * mwf) * NT + wave * 64 + (q & 31);
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    const int jr = f * MS + rg * (4 * NLG) + 4 * lg;
                    const int fbit = (((jr >> 5) * np + (q >> 5)) * 4 + ((jr & 31) >> 3)) * 4;
                    const uint32_t word = src[(fbit >> 5) * NT + (((jr & 31) >> 2) & 1) * 32];
                    const uint32_t nib = (word >> (fbit & 31)) & 0xFu;
                    const int bit = ((f * NP + 0) * RG + rg) * 4;
                    mw[bit >> 5] |= nib << (bit & 31);
                }
        } else {
#pragma unroll
            for (int w = 0; w < MW; ++w) mw[w] = masks[((l - 1) * MW + w) * NT + tid];
        }
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int j0 = feat0(f, rg);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int pt = p * MS + lp;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = j0 + i;
                        const int bit = ((f * NP + p) * RG + rg) * 4 + i;
                        float x = value(f, p, rg, i, k, pt);
                        if (k < prev_out) {
                            x = ((mw[bit >> 5] >> (bit & 31)) & 1u) ? x : 0.f;
                        } else {
                            if (k < inj_hi && slots[pt] >= 0)
                                atomicAdd(P.J + (int64_t)slots[pt] * NI + L.inj_off + (k - prev_out), x);
                            x = 0.f;
                        }
                        v[i] = x;
                    }
                    store4(act_e + ((j0 / KV) * PT + pt) * KV + (j0 % KV), v);
                }
            }
    };

    // top: in-gradient of the last linear = w_last[k] * gy[pt]
    store_in_grad(P.n_mfma, [&](int, int, int, int, int k, int pt) { return P.w_last[k] * gy[pt]; });
    __syncthreads();
    for (int l = P.n_mfma - 1; l >= 0; --l) {
        const MlpLayer L = P.L[l];
        gemm(reinterpret_cast<const vec_t*>(P.Wb) + L.off_b, L.kp_b, L.in_dim);
        __syncthreads();
        if (l > 0) {
            store_in_grad(l, [&](int f, int p, int rg, int i, int, int) { return acc[f][p][rg * 4 + i]; });
            __syncthreads();
        } else {
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    const int j0 = feat0(f, rg);
                    if (j0 >= NI) continue;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int pt = p * MS + lp;
                        if (slots[pt] < 0) continue;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (j0 + i < NI) atomicAdd(P.J + (int64_t)slots[pt] * NI + j0 + i, acc[f][p][rg * 4 + i]);
                    }
                }
        }
    }
    }   // JAC
}

// ---------------------------------------------------------------------------------------------------------------
// Host side: packing + C ABI
// ---------------------------------------------------------------------------------------------------------------

extern "C" int sdfr_decoder_create(sdfr_decoder** out, int n_lin, const int* in_dim, const int* out_dim,
                                   const int* inj_n, const int* inj_off, const float* const* h_W,
                                   const float* const* h_b, int n_inputs, int use_tanh, int device) {
    SDFR_REQUIRE(out && in_dim && out_dim && inj_n && inj_off && h_W && h_b, "sdfr_decoder_create: NULL argument");
    SDFR_REQUIRE(n_lin >= 2 && n_lin <= SDFR_MAX_LAYERS, "sdfr_decoder_create: n_lin=%d outside [2,%d]", n_lin, SDFR_MAX_LAYERS);
    SDFR_REQUIRE(out_dim[n_lin - 1] == 1, "sdfr_decoder_create: last layer must have out_dim 1 (got %d)", out_dim[n_lin - 1]);
    SDFR_REQUIRE(in_dim[0] == n_inputs && inj_n[0] == 0, "sdfr_decoder_create: layer 0 must consume exactly the input row");
    int width = 0;
    for (int l = 0; l < n_lin; ++l) {
        SDFR_REQUIRE(in_dim[l] > 0 && out_dim[l] > 0 && inj_n[l] >= 0 && inj_off[l] >= 0 && inj_off[l] + inj_n[l] <= n_inputs,
                     "sdfr_decoder_create: bad dims at layer %d", l);
        if (l > 0) SDFR_REQUIRE(in_dim[l] == out_dim[l - 1] + inj_n[l], "sdfr_decoder_create: layer %d in_dim %d != %d + %d", l,
                                in_dim[l], out_dim[l - 1], inj_n[l]);
        width = in_dim[l] > width ? in_dim[l] : width;
        if (l < n_lin - 1) width = out_dim[l] > width ? out_dim[l] : width;
    }
    SDFR_REQUIRE(width <= 512, "sdfr_decoder_create: hidden width %d > 512 unsupported", width);
    const int HP = width <= 128 ? 128 : (width <= 256 ? 256 : 512);
    SDFR_HIP_CHECK(hipSetDevice(device));

    sdfr_decoder* d = new sdfr_decoder();
    memset(d, 0, sizeof(*d));
    d->device = device; d->n_lin = n_lin; d->n_inputs = n_inputs; d->use_tanh = use_tanh; d->HP = HP;
    MlpParams& P = d->proto;
    P.n_mfma = n_lin - 1; P.n_inputs = n_inputs; P.use_tanh = use_tanh;
    int64_t off_f = 0, off_b = 0, off_h = 0;
    for (int l = 0; l < n_lin; ++l) {
        MlpLayer& L = P.L[l];
        L.in_dim = in_dim[l]; L.out_dim = out_dim[l]; L.inj_n = inj_n[l]; L.inj_off = inj_off[l];
        L.kp_f = 16 * ((in_dim[l] + 15) / 16); L.kp_b = 16 * ((out_dim[l] + 15) / 16); L.kp_h = 32 * ((in_dim[l] + 31) / 32);
        L.off_f = (int)off_f; L.off_b = (int)off_b; L.off_h = (int)off_h;
        d->macs += (int64_t)in_dim[l] * out_dim[l];
        if (l < n_lin - 1) { off_f += (int64_t)(L.kp_f / 4) * HP; off_b += (int64_t)(L.kp_b / 4) * HP; off_h += (int64_t)(L.kp_h / 8) * HP; }
    }
    // images: vector index [k / KV][row], KV consecutive k per 16-byte vector (KV = 4 floats or 8 halfs); zero padded
    std::vector<float> Wf((size_t)off_f * 4, 0.f), Wb((size_t)off_b * 4, 0.f), bias((size_t)(n_lin - 1) * HP, 0.f), wl(HP, 0.f);
    std::vector<_Float16> Wh((size_t)off_h * 8, (_Float16)0.f);
    for (int l = 0; l < n_lin - 1; ++l) {
        const MlpLayer& L = P.L[l];
        const float* W = h_W[l];
        for (int r = 0; r < L.out_dim; ++r)
            for (int k = 0; k < L.in_dim; ++k) {
                const float w = W[(size_t)r * L.in_dim + k];
                Wf[((size_t)L.off_f + (size_t)(k / 4) * HP + r) * 4 + (k % 4)] = w;
                Wh[((size_t)L.off_h + (size_t)(k / 8) * HP + r) * 8 + (k % 8)] = (_Float16)w;
                Wb[((size_t)L.off_b + (size_t)(r / 4) * HP + k) * 4 + (r % 4)] = w;          // transposed: rows = in-features, k = out-features
            }
        for (int r = 0; r < L.out_dim; ++r) bias[(size_t)l * HP + r] = h_b[l][r];
    }
    for (int k = 0; k < in_dim[n_lin - 1]; ++k) wl[k] = h_W[n_lin - 1][k];
    P.b_last = h_b[n_lin - 1][0];

    SDFR_HIP_CHECK(hipMalloc(&d->d_Wf, Wf.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMalloc(&d->d_Wb, Wb.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMalloc(&d->d_Wh, Wh.size() * sizeof(_Float16)));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wh, Wh.data(), Wh.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMalloc(&d->d_bias, bias.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMalloc(&d->d_wlast, wl.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wf, Wf.data(), Wf.size() * sizeof(float), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wb, Wb.data(), Wb.size() * sizeof(float), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMemcpy(d->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMemcpy(d->d_wlast, wl.data(), wl.size() * sizeof(float), hipMemcpyHostToDevice));
    P.Wf = d->d_Wf; P.Wb = d->d_Wb; P.Wh = d->d_Wh; P.bias = d->d_bias; P.w_last = d->d_wlast; P.fwd_np = 2;
    *out = d;
    return SDFR_OK;
}

extern "C" int sdfr_decoder_destroy(sdfr_decoder* d) {
    if (!d) return SDFR_OK;
    hipFree(d->d_Wf); hipFree(d->d_Wb); hipFree(d->d_Wh); hipFree(d->d_bias); hipFree(d->d_wlast);
    delete d;
    return SDFR_OK;
}

extern "C" int64_t sdfr_decoder_macs(const sdfr_decoder* d) { return d ? d->macs : 0; }

// per-thread mask words of the forward kernel for a padded width HP: FT (= feature tiles per wave), threads per workgroup NT
static void mask_geometry(int HP, int* ft, int* nt) {
    if (HP == 128) { *ft = 1; *nt = 256; }
    else if (HP == 256) { *ft = 2; *nt = 256; }
    else { *ft = 2; *nt = 512; }
}

extern "C" int64_t sdfr_decoder_mask_words(const sdfr_decoder* d, int64_t n) {
    if (!d || n <= 0) return 0;
    int ft, nt;
    mask_geometry(d->HP, &ft, &nt);
    // 512 bits per point and layer for HP = 512, rounded up to whole 128-point tiles (covers the f32 64-point and f16 128-point layouts)
    return ((n + 127) / 128) * 2 * (int64_t)(d->n_lin - 1) * ft * nt;
}

extern "C" int sdfr_mlp_forward(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf, "sdfr_mlp_forward: NULL argument");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward: n=%lld out of range", (long long)n);
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = mask_ws;
    hipStream_t s = (hipStream_t)stream;
    const int grid = sdfr_cdiv(n, 64);
    static const int variant = getenv("SDFR_MLP_VARIANT") ? atoi(getenv("SDFR_MLP_VARIANT")) : 0;   // development A/B switch
    if (mask_ws) {
        switch (d->HP) {
            case 128: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 2, 4, 2, 1>), dim3(grid), dim3(256), 0, s, P); break;
            case 256: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 4, 2, 1>), dim3(grid), dim3(256), 0, s, P); break;
            default:  hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, 4, 1>), dim3(grid), dim3(512), 0, s, P); break;
        }
    } else {
        switch (d->HP) {
            case 128: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 2, 4, 2, 0>), dim3(grid), dim3(256), 0, s, P); break;
            case 256: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 4, 2, 0>), dim3(grid), dim3(256), 0, s, P); break;
            default:
                if (variant == 7) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 4, 2, 4, 2, 0>), dim3(grid), dim3(256), 0, s, P);
                else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, 4, 0>), dim3(grid), dim3(512), 0, s, P);
                break;
        }
    }
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// forward with float16 operands (f32 accumulate, f32 bias/ReLU/tanh): 128-point workgroup tiles
extern "C" int sdfr_mlp_forward_f16(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf, "sdfr_mlp_forward_f16: NULL argument");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward_f16: n=%lld out of range", (long long)n);
    SDFR_REQUIRE(d->HP == 512, "sdfr_mlp_forward_f16: built for hidden widths 257..512 (padded width %d)", d->HP);
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = mask_ws;
    hipStream_t s = (hipStream_t)stream;
    const int grid = sdfr_cdiv(n, 128);
    if (mask_ws) hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 1>), dim3(grid), dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 0>), dim3(grid), dim3(512), 0, s, P);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_mlp_jacobian(const sdfr_decoder* d, const float* inputs, int64_t rows_per_crop, int B,
                                 const int32_t* idx, int cap, const int32_t* cnt, float* J, float* sdf_sel,
                                 const float* sdf_full, const uint32_t* mask_ws, int mask_from_f16, void* stream) {
    SDFR_REQUIRE(d && inputs && idx && J, "sdfr_mlp_jacobian: NULL argument");
    SDFR_REQUIRE(B >= 0 && cap >= 0, "sdfr_mlp_jacobian: negative size");
    SDFR_REQUIRE((mask_ws == nullptr) == (sdf_full == nullptr) || mask_ws == nullptr, "sdfr_mlp_jacobian: mask_ws needs sdf_full");
    if (B == 0 || cap == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    SDFR_HIP_CHECK(hipMemsetAsync(J, 0, (size_t)B * cap * d->n_inputs * sizeof(float), s));
    MlpParams P = d->proto;
    P.inputs = inputs; P.rows_per_crop = rows_per_crop; P.idx = idx; P.cnt = cnt; P.cap = cap; P.J = J; P.sdf_sel = sdf_sel;
    P.sdf_in = sdf_full; P.maskbuf = const_cast<uint32_t*>(mask_ws); P.fwd_np = mask_from_f16 ? 4 : 2;
    dim3 grid(sdfr_cdiv(cap, 32), B), grid16(sdfr_cdiv(cap, 16), B);
    static const int jvar = getenv("SDFR_JAC_VARIANT") ? atoi(getenv("SDFR_JAC_VARIANT")) : 0;   // development A/B switch
    // masks saved by the forward launch make the recomputation unnecessary (not for use_tanh decoders: their output
    // derivative needs the pre-tanh value)
    if (mask_ws && sdf_full && !d->use_tanh) {
        switch (d->HP) {
            case 128: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 1, 4, 2, 3>), grid, dim3(256), 0, s, P); break;
            case 256: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 4, 2, 3>), grid, dim3(256), 0, s, P); break;
            default:
                if (jvar == 1) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 8, 4, 3>), grid, dim3(512), 0, s, P);
                else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, 4, 1, 8, 4, 3>), grid16, dim3(512), 0, s, P);
                break;
        }
    } else {
        switch (d->HP) {
            case 128: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 1, 4, 2, 2>), grid, dim3(256), 0, s, P); break;
            case 256: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 4, 2, 2>), grid, dim3(256), 0, s, P); break;
            default:
                if (jvar == 1) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 8, 4, 2>), grid, dim3(512), 0, s, P);
                else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, 4, 1, 8, 4, 2>), grid16, dim3(512), 0, s, P);
                break;
        }
    }
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
