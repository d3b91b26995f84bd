// DeepSDF decoder on MI355X (gfx950): fused multi-layer MLP forward and input-Jacobian backward.
//
// Replaces Decoder.forward (reference sdfrenderer/deepsdf/networks/deep_sdf_decoder_scale.py:78-107) and the
// autograd backward of it w.r.t. its input rows (the normals hook of sdfrenderer/grid.py:55-56 and the latent
// gradient of pipelines/optimizer.py:156).
//
// Design (CDNA4-first, nothing here is translated from the reference's ATen graph):
//   * One workgroup (8 waves, two per SIMD, for the 512-wide decoder) owns a tile of PT = MS*NP grid points and carries them
//     through EVERY layer.  Activations never leave the CU: they live in LDS as  act[k/KV][point][k%KV]  (16-byte vectors),
//     128 KiB for 512 features x 64 points (float) or x 128 points (half).
//   * Each layer is computed transposed,  out^T[feature][point] = W[feature][k] * act^T[k][point],  with the exact-f32 matrix
//     instruction v_mfma_f32_32x32x2_f32 (or v_mfma_f32_32x32x16_f16 with f32 accumulation).  Wave w owns output features
//     [w*MS*FT, (w+1)*MS*FT): FT x NP accumulator tiles.  The MFMA A operand (weights) is NOT shared between waves, so it is streamed
//     straight from L2 into VGPRs (no LDS staging, no barrier in the K loop) from an image packed once at load time,
//     W[k/KV][row][k%KV]: each lane's fragment is one coalesced 16-byte load.  The B operand (activations) is one conflict-free
//     ds_read_b128 per MS points.  With the transposed product a lane's 4 consecutive accumulator registers are 4 consecutive
//     features of ONE point, so the epilogue (bias + ReLU + latent re-injection) writes the next layer's operand with conflict-free
//     wide LDS stores.  Only two barriers per layer.
//   * The last linear (H -> 1) is a VALU dot product out of LDS followed by tanh.
//   * Jacobian modes: the ReLU masks (1 bit per feature per point per layer; saved to HBM by the grid forward, or rebuilt in LDS by
//     recomputation) are all the backward needs, because only the INPUT gradient is required (weights are frozen); the layers run
//     backwards with the transposed weight image Wb, giving d sdf / d input-row for the selected rows only.
#pragma once
#include "sdfr_common.h"
#include <math.h>
#include <type_traits>

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
    int kp_bh, off_bh;      // float16 backward: out_dim padded to x128, 16-byte-vector offset of the layer's image in Wbh
    int kp_s, off_s;        // split forward: K padded to x64, 16-byte-vector offset of the layer's image in Ws
    int ln;                 // LayerNorm (eps 1e-5, affine) between this layer's linear and its ReLU (deep_sdf_decoder_scale.py:56-57,99-101)
};

struct MlpParams {
    const float4* Wf;       // forward image, float32:  [k/4][HP] float4
    const float4* Wb;       // backward (transposed) image, float32
    const void* Wh;         // forward image, float16:  [k/8][HP] 8 x half
    const void* Wbh;        // backward (transposed) image, float16:  [j/8][HP] 8 x half (the float16 decoder's mask-fed Jacobian)
    const void* Ws;         // split-forward image: [k/8][HP][2] 8 x half -- hi = half(w) and lo = half((w - hi) * 2^11) side by side
    int kinj;               // half forward image Wh: re-injected input columns sit at k = HP + input column (K-side injection, n_inputs <= 8)
    int fwd_np;             // MODE 3: point tiles per workgroup of the forward launch that saved the masks (2: f32, 4: f16)
    const float* bias;      // [n_mfma][HP]
    const float* ln_gamma;  // [n_mfma][HP] LayerNorm weight / bias (LN decoders only)
    const float* ln_beta;
    float4* ln_ws;          // LN Jacobian scratch: normalised pre-activations [workgroup][layer][HP/4][PT] float4
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
    const int32_t* skip;         // forward: optional per-crop flags (skip_rows rows per crop): workgroups whose rows all belong to flagged crops exit
    int64_t skip_rows;
    const int32_t* crop_cnt;     // forward: optional per-crop row counts of a [B][crop_rows] row array (crop_rows a multiple of the tile): tiles that
    int64_t crop_rows;           // start at or beyond their crop's count exit (candidate rows of the float16 reuse mode, r05)
    const int32_t* gather_idx;   // GATHER kernels (r06): row s of crop c of the ragged [B][crop_rows] launch is inputs[c * gather_rows + gather_idx[c * crop_rows + s]]
    int64_t gather_rows;         // (s < crop_cnt[c]; rows beyond the count read the crop's row 0) -- the candidate rows are read where they lie, no copy
    int n_crops;                 // PERSIST kernels (r06): crops of the ragged / skip launch (the workgroups walk the LIVE tiles only)
    const int32_t* n_dev;        // forward: optional device-side row count (rows >= *n_dev are not evaluated; n is the launch bound)
    int n_dev_lo, n_dev_hi;      // with n_dev and n_dev_hi > 0: the launch runs only while n_dev_lo <= *n_dev < n_dev_hi (two tile geometries of one step)
    unsigned long long* trace;   // builds with -DSDFR_MLP_TRACE: cycle stamps of workgroup 0 (sdfr_debug_set_trace), else unused
    // MODE 4 (sphere tracing, persistent tail: csrc/trace.hip): the workgroup marches the t_rt rays of its tile by itself -- decoder pass on
    // PT rows (up to PT / t_rt samples per ray), step rule, hit / exit test, next pass -- rewriting its own operand rows between passes, until
    // the rays have terminated, the step budget is spent, or the stage ends (t_stage passes: the survivors are appended to the next stage's list)
    float* t_rows;               // = inputs: scratch [tile][PT][n_inputs], row j*t_rt + i = sample j of the tile's ray i
    int t_rt;                    // rays per tile (16, 8 or 4; PT / t_rt = most samples per ray and pass this launch can carry)
    const float* t_latn;         // [B][L] normalised latents
    const int32_t* t_pix;        // active list: crop * W*H + pixel
    const float4* t_lam;         // active list: ray state (lam = next sample, rho = |sdf| of the previous accepted sample, q = ratio of the last two radii, -)
    int t_step0;                 // pass index of the kernel's first pass
    int t_nlv;                   // speculation schedule: passes with index >= t_lv_from[i] take t_lv_k[i] samples per ray (levels ascending; before the
    int t_lv_from[SDFR_TRACE_LEVELS], t_lv_k[SDFR_TRACE_LEVELS];   // first level: 1 sample); the pass INDEX alone decides, never a count
    float t_qmax;                // upper clamp of the radius ratio q that spaces the speculative samples (lower clamp 0.5)
    int32_t* t_tile_ctr;         // device counter (zeroed by sdfr_trace_setup) the workgroups of this launch fetch their tiles from: the grid is a
                                 // bounded pool of persistent workgroups, each with ONE tile of scratch rows, whatever the ray count
    int t_stage;                 // passes this launch may run (<= t_steps)
    int32_t* t_next_cnt;         // next stage's active list (NULL: none): count, pixels, states
    int32_t* t_next_pix;
    float4* t_next_lam;
    const float* t_far;          // per pixel: ray parameter at which the ray leaves the object cube
    const float* t_pose;         // [B][16]
    const float* t_Kinv;         // [B][9]
    float* t_hit_lam;            // per pixel: ray parameter of the hit (0: none)
    float* t_hit_sdf;            // per pixel: decoder value at the marched hit
    int t_W, t_H, t_steps;       // image size; passes left in the march's step budget
    const int32_t* t_wh; int t_PS;  // ragged extents (r04): per-crop (W_b, H_b) on the device or NULL; pixel slot per crop (W H when dense)
    float t_eps, t_sigma;
    unsigned long long* t_evals; // += active rays per pass (ray evaluations of the march, for the roofline)
    int32_t* t_unresolved;       // += rays still active when the step budget ran out
};

struct sdfr_decoder {
    int device;
    int n_lin, n_inputs, use_tanh, HP;
    float4* d_Wf;
    float4* d_Wb;
    void* d_Wh;
    void* d_Ws;
    void* d_Wbh;
    float* d_lng;           // LayerNorm weight / bias images [n_mfma][HP] (NULL without LN)
    float* d_lnb;
    int has_ln;
    mutable float4* ln_ws;  // Jacobian scratch of LN decoders, grown on demand
    mutable size_t ln_ws_bytes;
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
#ifdef SDFR_ABL_NOMFMA
        c[0] += (float)a[0] * (float)b[0];       // ablation (timing only, wrong results): the loads stay live, the matrix pipe stays idle
        return c;
#else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
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

// ReLU masks in HBM (layout v2, r06): one bit per (row, layer, feature), independent of the tile geometry of the launch that saved them.  Rows
// in blocks of 128; a row's HP bits of one layer are HP / 32 consecutive dwords (64 bytes for HP = 512), a block's rows of one layer are
// contiguous (8 KiB), then the next layer:   dword (((r >> 7) * n_layers + l) * 128 + (r & 127)) * (HP / 32) + (j >> 5)   holds features
// j & ~31 ... of row r, layer l, feature j at bit ((j >> 2) & 1) * 16 + ((j & 31) >> 3) * 4 + (j & 3) -- the order in which a thread of the 32x32
// forward kernels holds them (lane group lg = (j >> 2) & 1 owns one 16-bit half: a thread stores its 16 bits per (feature tile, point) as one
// short; the workgroup's stores fill whole 64-byte rows).  r05's layout was [forward tile][layer][word][thread]: a band row's 64 bytes per
// layer were 32 dwords 128 bytes apart, at places that depended on the forward launch's tile size.
__device__ __forceinline__ int64_t sdfr_mask_dword(int64_t r, int l, int n_layers, int HP32, int j) {
    return (((r >> 7) * n_layers + l) * 128 + (r & 127)) * HP32 + (j >> 5);
}
__device__ __forceinline__ int sdfr_mask_shift(int j) { return ((j >> 2) & 1) * 16 + ((j & 31) >> 3) * 4 + (j & 3); }

// ET   operand element type (float: exact f32; _Float16: half operands, f32 accumulate -- forward modes and the mask-fed Jacobian)
// MS   MFMA tile (32: forward on the grid; 16: small tiles so that a few thousand band rows fill the chip)
// FT   feature tiles (MS rows) per wave, NP point tiles (MS points) per workgroup, NW waves per workgroup (HP = MS*FT*NW padded width)
// PF   weight/activation fragment buffers in flight per wave (prefetch distance PF-1 K tiles)
// MODE 0: forward.  1: forward + ReLU masks saved to HBM (1 bit per feature, point, layer).  2: Jacobian of selected rows by
//      recomputation (forward with masks in LDS, then backward).  3: Jacobian of selected rows from the masks a MODE-1 launch saved
//      (backward only: no activations are needed for an input gradient, only the masks and the output).
// Lane map of one MS x MS accumulator tile: point = lane % MS, feature = (reg/4)*4*NLG + 4*(lane/MS) + reg%4 with NLG = 64/MS lane
// groups; a lane's 4 consecutive registers are 4 consecutive features of one point.
// LN   decoder variant with LayerNorm after the hidden linears (weight_norm=False, norm_layers): forward and recomputing Jacobian only
#ifndef SDFR_MLP_WPE
#define SDFR_MLP_WPE 1      // minimum waves per SIMD the register allocation must allow (A/B builds: 2 = two 4-wave workgroups per CU)
#endif
// XF   extra forward features (r06; separate instantiations in mlp_persist.hip, the hot kernels are compiled without them):
//      1 GATHER   the tile's rows are read through an index list (P.gather_idx: candidate rows of the reuse modes, no gathered copy)
//      2 PERSIST  the launch is a fixed pool of workgroups that walk the LIVE tiles of a ragged [B][crop_rows] launch (per-crop counts) or of a
//                 skip launch (per-crop flags): dead tiles cost nothing, a launch with nothing to do costs one wave of dispatch, and the last
//                 round of a many-crop launch is as full as the live tile count allows
template <typename ET, int MS, int FT, int NP, int NW, int PF, int MODE, int PFB_ = 0, bool LN = false, int XF = 0>
__global__ __launch_bounds__(64 * NW, SDFR_MLP_WPE) void sdfr_mlp_kernel(const MlpParams P) {
    typedef Mma<ET, MS> M;
    typedef typename M::acc_t acc_t;
    typedef typename M::vec_t vec_t;
    constexpr bool HALF = sizeof(ET) == 2;
    constexpr bool JAC = MODE == 2 || MODE == 3;
    constexpr bool TAIL = MODE == 4;                               // forward passes in a loop: the sphere tracer's persistent tail
    constexpr bool SAVE = MODE == 1;
    constexpr bool LMASK = MODE == 2;
    constexpr bool GMASK = MODE == 3;
    constexpr bool GATHER = (XF & 1) != 0;
    constexpr bool PERSIST = (XF & 2) != 0;
    static_assert(XF == 0 || (MODE == 0 || MODE == 1) || (MODE == 3 && XF == 2), "GATHER is a forward feature; PERSIST: forward modes and the mask-fed Jacobian");
    constexpr bool JPOOL = JAC && PERSIST;                         // the mask-fed Jacobian as a pool over the crops' live band tiles (r06)
    static_assert(!SAVE || MS == 32, "mask layout assumes 32x32 forward tiles");
    static_assert(!SAVE || (FT * NP) % 2 == 0, "mask words: FT*NP*16 bits per thread and layer must fill whole words");
    static_assert(!HALF || !JAC || MODE == 3, "with half operands only the mask-fed Jacobian exists (MODE 3)");
    static_assert(!LN || (!HALF && MODE != 1 && MODE != 3 && MODE != 4), "LayerNorm decoders: float32 forward (MODE 0) and recomputing Jacobian (MODE 2)");
    static_assert(!TAIL || (NP * MS <= 64 && (NP * MS) % 16 == 0), "tail march: the tile's rows (t_rt rays x K samples) live in wave 0");
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
    constexpr int LN_F4 = LN ? (NW * PT + SDFR_MAX_LAYERS * PT + 3) / 4 : 0;   // cross-wave partial sums + rstd per layer
    // half forward: four more k groups behind the HP feature slots hold the tile's input rows (k = HP + input column; zero beyond), so that
    // layers which re-inject input columns (latent_in / xyz_in_all) find them in the operand at a fixed place and no epilogue has to patch
    // them in (the generic per-element injection path cost the one wave that ran it 12 k cycles per layer, everybody else waiting)
    constexpr int KGX = KG + ((HALF && !JAC) ? 4 : 0);
    // Two operand tiles in turn where LDS has the room (half forward modes on tiles of up to 64 rows: 2 x 70 KB): a layer's epilogue writes the
    // NEXT operand into the other tile, so no wave has to wait until every wave has finished READING the current one -- one barrier per layer
    // instead of two.  These tiles are the sphere tracer's passes, each a chain of 8 latency-bound layers (r04, profiles/r04_notes.md section 8).
#ifndef SDFR_ACT_DBUF
#define SDFR_ACT_DBUF 1
#endif
    constexpr bool DBUF = SDFR_ACT_DBUF && HALF && !JAC && !LN && !SAVE && (2 * KGX * PT * 16 <= 144 * 1024);
    constexpr int ACT4 = (DBUF ? 2 : 1) * KGX * PT;
    // MLDS (r06; the half mask-fed Jacobian on 16x16 products): the ReLU masks of EVERY layer for the tile's rows are fetched once, in the
    // kernel's prologue, and kept in LDS.  Fetched layer by layer in front of each product (r05) the 16 scattered 4-byte gathers per lane sat
    // in front of the layer's first weight fragment in the in-order vmcnt queue (one exposed HBM / L2-miss latency per layer), and unpacking
    // them took 530 VALU instructions per wave and layer.
    constexpr bool MLDS = GMASK && HALF && MS == 16 && HP == 512;
    static_assert(!MLDS || (FT == 4 && NW == 8), "MLDS: a wave owns 64 features = two mask dwords of a row");
    constexpr int MLDS_WORDS = MLDS ? NW * SDFR_MAX_LAYERS * PT * 2 : 0;   // [wave][layer][row] 8 bytes: the wave's 64 mask bits of the row
    __shared__ float4 lds4[ACT4 + NT / 4 + 32 + (PT + 3) / 4 * 2 + (MASK_WORDS + 3) / 4 + LN_F4 + (MLDS_WORDS + 3) / 4];
    vec_t* act = reinterpret_cast<vec_t*>(lds4);                  // [KG][PT] 16-byte vectors: act[k/KV][point][k%KV] -- the operand the products read
    ET* act_e = reinterpret_cast<ET*>(lds4);
    vec_t* actw = DBUF ? act + KGX * PT : act;                    // ... and the one the epilogue writes (the same without the second tile)
    ET* actw_e = reinterpret_cast<ET*>(actw);
    float* red = reinterpret_cast<float*>(lds4 + ACT4);            // [NT]
    int* rows = reinterpret_cast<int*>(lds4 + ACT4 + NT / 4);      // [PT] source row of each point (128 ints max)
    float* gy = reinterpret_cast<float*>(lds4 + ACT4 + NT / 4 + 32);   // [PT] d out / d y_last
    int* slots = reinterpret_cast<int*>(lds4 + ACT4 + NT / 4 + 32 + (PT + 3) / 4);   // [PT] J slot or -1
    uint32_t* masks = reinterpret_cast<uint32_t*>(lds4 + ACT4 + NT / 4 + 32 + (PT + 3) / 4 * 2);
    float* lnred = reinterpret_cast<float*>(lds4 + ACT4 + NT / 4 + 32 + (PT + 3) / 4 * 2 + (MASK_WORDS + 3) / 4);   // [NW][PT]
    float* lnrstd = lnred + NW * PT;                                                                                   // [layers][PT]
    uint32_t* mlds = reinterpret_cast<uint32_t*>(lds4 + ACT4 + NT / 4 + 32 + (PT + 3) / 4 * 2 + (MASK_WORDS + 3) / 4 + LN_F4);   // uint2 [NW][layers][PT]
    // MLDS kernels on decoders with at most 8 input columns: a J row is assembled in LDS and written ONCE -- gradients of re-injected input
    // columns (latent_in / xyz_in_all layers) are added here, the first layer's in-gradient joins them at the end.  (r05 zeroed the rows in
    // HBM at the start and atomicAdd-ed both parts: 24 lane-divergent global atomics in the last wave of the re-injecting layer held the
    // other seven waves at the barrier for 4-7 k cycles.)
    __shared__ float jinj[MLDS ? PT * 8 : 1];
    __shared__ int jpfx[JPOOL ? 66 : 1];                          // JPOOL: inclusive prefix of the crops' live tiles (B <= 64), [64] = total

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lp = lane % MS;              // point within a point tile
    const int lg = lane / MS;              // lane group: k slot of the operands, feature sub-block of the accumulator
    const int NI = P.n_inputs;

#ifdef SDFR_MLP_TRACE
    // cycle stamps of workgroup 0, waves 0 and NW-1, lane 0:  trace[(wave ? 1 : 0)][layer][5] = layer start, product loop done, first barrier
    // passed, epilogue done, second barrier passed (s_memtime ticks = shader cycles)
#define SDFR_STAMP(l, i)                                                                                                   \
    do {                                                                                                                   \
        if (P.trace && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == NW - 1))                   \
            P.trace[((wave ? 1 : 0) * SDFR_MAX_LAYERS + (l)) * 5 + (i)] = __builtin_amdgcn_s_memtime();                    \
    } while (0)
#else
#define SDFR_STAMP(l, i) do { } while (0)
#endif
    SDFR_STAMP(8, 0);
    // ---- which rows does this tile hold ------------------------------------------------------------------
    // MODE 4: a pool of persistent workgroups; each trip of this loop fetches one tile (t_rt rays) from the launch's device counter and marches
    // it through the stage.  Every other mode: one trip, tile = blockIdx.x.
    int* tile_slot = reinterpret_cast<int*>(gy) + 1;
    // PERSIST: live tiles of the launch, counted per crop.  Ragged launch (crop_cnt): crop c has ceil(min(cnt[c], crop_rows) / PT) live tiles
    // at the head of its crop_rows / PT tile slots; skip launch (skip flags; crop_rows == 0): a crop is live or not as a whole, and the walk
    // is over all tiles of the launch (a tile may span crops).  pfx[c] = live tiles of the crops before c (inclusive scan in slots[], which
    // forward modes do not use otherwise; B <= PT entries -- larger launches fall back to walking every tile slot).
    int p_iter = 0, p_live = 0;
    bool p_compact = false;
    if constexpr (PERSIST) {
        if (P.crop_cnt && P.n_crops <= PT) {
            if (tid == 0) {
                int acc_ = 0;
                for (int c = 0; c < P.n_crops; ++c) {
                    const int64_t cc = min((int64_t)P.crop_cnt[c], P.crop_rows);
                    acc_ += (int)((cc + PT - 1) / PT);
                    slots[c] = acc_;
                }
                tile_slot[1] = acc_;
            }
            __syncthreads();
            p_live = tile_slot[1];
            p_compact = true;
            if ((int)blockIdx.x >= p_live) return;
        } else if (P.skip && !P.crop_cnt) {
            bool any = false;
            for (int c = 0; c < P.n_crops; ++c) any = any || P.skip[c] == 0;
            if (!any) return;                               // nothing to evaluate in this launch: the common step of a reuse refinement
        }
        if constexpr (JPOOL) {
            // live tiles of crop c: ceil(min(cnt[c], cap) / PT); the pool walks them back to back (a workgroup per CU stays resident: no
            // dispatch gap between a CU's tiles, and the last round is as full as the live tile count allows)
            if (tid == 0) {
                int acc_ = 0;
                for (int c = 0; c < P.n_crops; ++c) { acc_ += (sdfr_count(P.cnt, c, P.cap) + PT - 1) / PT; jpfx[c] = acc_; }
                jpfx[64] = acc_;
            }
            __syncthreads();
            p_live = jpfx[64];
            if ((int)blockIdx.x >= p_live) return;
        }
    }
    do {
    int tile = blockIdx.x;
    if constexpr (PERSIST) {
        const int64_t j = (int64_t)blockIdx.x + (int64_t)p_iter * gridDim.x;
        ++p_iter;
        if constexpr (JPOOL) {
            if (j >= p_live) return;
            tile = (int)j;                                  // (mapped to crop and band tile below)
        } else
        if (p_compact) {
            if (j >= p_live) return;
            int lo = 0, hi = P.n_crops - 1;                 // first crop whose inclusive prefix exceeds j
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (slots[mid] > (int)j) hi = mid; else lo = mid + 1; }
            const int before = lo > 0 ? slots[lo - 1] : 0;
            tile = (int)((int64_t)lo * (P.crop_rows / PT) + (j - before));
        } else {
            if (j * PT >= P.n) return;
            tile = (int)j;
        }
        if (p_iter > 1) __syncthreads();                    // the previous tile's readers of rows[] / the operand tile are through
    }
    if constexpr (TAIL) {
        if (P.n_dev && P.n_dev_hi > 0 && (*P.n_dev < P.n_dev_lo || *P.n_dev >= P.n_dev_hi)) return;      // gated launch: nothing fetched
        if (P.t_steps <= 0) return;
        if (P.t_tile_ctr) {
            __syncthreads();                                   // the previous tile's readers of the LDS state are through
            if (tid == 0) tile_slot[0] = atomicAdd(P.t_tile_ctr, 1);
            __syncthreads();
            tile = tile_slot[0];
        }
    }
    int n_valid;
    if (JAC) {
        int b = blockIdx.y, s0 = blockIdx.x * PT;
        if constexpr (JPOOL) {
            int lo = 0, hi = P.n_crops - 1;                 // first crop whose inclusive prefix exceeds the tile index
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (jpfx[mid] > tile) hi = mid; else lo = mid + 1; }
            b = lo;
            s0 = (tile - (lo > 0 ? jpfx[lo - 1] : 0)) * PT;
        }
        const int count = sdfr_count(P.cnt, b, P.cap);
        if (s0 >= count) { if constexpr (JPOOL) continue; else return; }
        n_valid = min(PT, count - s0);
        if (tid < PT) {
            const bool v = tid < n_valid;
            const int s = v ? (s0 + tid) : s0;
            rows[tid] = (int)(P.rows_per_crop * b) + P.idx[(int64_t)b * P.cap + s];
            slots[tid] = v ? (b * P.cap + s) : -1;
        }
    } else {
        // (MODE 4: a tile is t_rt RAYS of the active list -- n / n_dev count rays -- and PT operand rows of the tile's own scratch)
        const int64_t r0 = (int64_t)tile * (TAIL ? P.t_rt : PT);
        const int64_t n_rows = P.n_dev ? min(P.n, (int64_t)*P.n_dev) : P.n;          // sphere tracing: the active-ray count lives on the device
        if (r0 >= n_rows) return;
        if (P.n_dev && P.n_dev_hi > 0 && (*P.n_dev < P.n_dev_lo || *P.n_dev >= P.n_dev_hi)) return;
        if (TAIL && P.t_steps <= 0) return;
        if (P.crop_cnt) {                           // ragged [B][crop_rows] row array: nothing to do beyond the crop's own count
            const int64_t c = r0 / P.crop_rows;
            if (r0 - c * P.crop_rows >= (int64_t)P.crop_cnt[c]) { if constexpr (PERSIST) continue; else return; }
        }
        if (P.skip) {                               // two-stage evaluation: crops that reuse their candidate set skip the half pass
            // a tile is dropped only when EVERY crop it spans is flagged (rows_per_crop need not be a multiple of the tile: a tile may span
            // two or more crops); in a surviving tile the rows of flagged crops are computed but not stored (see the store of P.sdf)
            const int64_t r1 = min(r0 + PT, n_rows) - 1;
            bool all_flagged = true;
            for (int64_t c = r0 / P.skip_rows; c <= r1 / P.skip_rows; ++c) all_flagged = all_flagged && P.skip[c] != 0;
            if (all_flagged) { if constexpr (PERSIST) continue; else return; }
        }
        n_valid = (int)min((int64_t)(TAIL ? P.t_rt : PT), n_rows - r0);
        if constexpr (GATHER) {
            if (tid < PT) {
                const int64_t c = r0 / P.crop_rows, sc = r0 - c * P.crop_rows + tid;
                const bool live = sc < (int64_t)P.crop_cnt[c];          // beyond the count: the crop's row 0 (finite padding rows of the last tile)
                rows[tid] = (int)(c * P.gather_rows + (live ? P.gather_idx[c * P.crop_rows + sc] : 0));
            }
        } else
        if (tid < PT) rows[tid] = TAIL ? (int)((int64_t)blockIdx.x * PT + tid) : (int)(r0 + (tid < n_valid ? tid : 0));
    }
    // ---- MODE 4: the tile's rays.  Lane i < RT of wave 0 owns ray i: state and ray in registers ----
    const int RT = TAIL ? P.t_rt : 1;                   // rays of the tile
    const int KMAX = TAIL ? PT / RT : 1;                // most samples per ray and pass the tile has rows for
    // samples per ray of the pass with index s: the schedule, a function of s alone (clipped to what the tile can carry: the host picks t_rt so
    // that it never clips)
    auto pass_k = [&](int s) {
        int k = 1;
        for (int i = 0; i < P.t_nlv; ++i) k = (s >= P.t_lv_from[i]) ? P.t_lv_k[i] : k;
        return min(KMAX, k);
    };
    int t_gp = 0, t_left = 0, t_pass = 0;
    bool t_act = false;
    float4 t_st = make_float4(0.f, 0.f, 1.f, 0.f);
    float t_farl = 0.f, t_idn = 1.f, t_ox = 0.f, t_oy = 0.f, t_oz = 0.f, t_dx = 0.f, t_dy = 0.f, t_dz = 0.f;
    unsigned long long t_ev = 0ull;
    // sample positions of a pass with k live samples (p_0 = lam, p_j = p_{j-1} + sigma q^j rho / |d|; slots j >= k repeat p_0) -> operand rows
    // (the step rule below recomputes the positions with the same operations instead of keeping them in registers)
    auto tail_rows = [&](int k, bool with_latent) {
        if (tid >= RT) return;
        float pj = t_st.x, qp = t_st.z;
        for (int j = 0; j < KMAX; ++j) {
            if (j > 0 && j < k) { pj = pj + ((P.t_sigma * qp) * t_st.y) / t_idn; qp = qp * t_st.z; }
            const float pos = (j < k) ? pj : t_st.x;
            float* row = P.t_rows + ((int64_t)blockIdx.x * PT + j * RT + tid) * NI;
            if (with_latent) {
                const int P_ = P.t_PS;
                const float* lz = P.t_latn + (int64_t)(t_gp / P_) * (NI - 3);
                for (int c = 0; c < NI - 3; ++c) row[c] = lz[c];
            }
            row[NI - 3] = t_ox + pos * t_dx; row[NI - 2] = t_oy + pos * t_dy; row[NI - 1] = t_oz + pos * t_dz;
        }
    };
    if constexpr (TAIL) {
        t_left = min(P.t_steps, P.t_stage);
        if (tid < RT) {
            const int64_t s = (int64_t)tile * RT + (tid < n_valid ? tid : 0);              // lanes beyond the tile's rays mirror ray 0 (finite rows), inactive
            t_gp = P.t_pix[s];
            t_st = P.t_lam[s];
            t_farl = P.t_far[t_gp];
            const int P_ = P.t_PS, b = t_gp / P_, px = t_gp - b * P_;
            const int Wb = P.t_wh ? P.t_wh[2 * b] : P.t_W;
            const float* Pm = P.t_pose + (int64_t)b * 16;
            const float* Ki = P.t_Kinv + (int64_t)b * 9;
            const float x = (float)(px % Wb), y = (float)(px / Wb);
            const float rx = fmaf(Ki[1], y, Ki[0] * x) + Ki[2], ry = fmaf(Ki[4], y, Ki[3] * x) + Ki[5], rz = fmaf(Ki[7], y, Ki[6] * x) + Ki[8];
            t_dx = Pm[0] * rx + Pm[4] * ry + Pm[8] * rz;
            t_dy = Pm[1] * rx + Pm[5] * ry + Pm[9] * rz;
            t_dz = Pm[2] * rx + Pm[6] * ry + Pm[10] * rz;
            t_ox = -(Pm[0] * Pm[3] + Pm[4] * Pm[7] + Pm[8] * Pm[11]);
            t_oy = -(Pm[1] * Pm[3] + Pm[5] * Pm[7] + Pm[9] * Pm[11]);
            t_oz = -(Pm[2] * Pm[3] + Pm[6] * Pm[7] + Pm[10] * Pm[11]);
            t_idn = sqrtf(t_dx * t_dx + t_dy * t_dy + t_dz * t_dz);        // |d| (the step divides by it, as sdfr_trace_step_kernel does)
            t_act = tid < n_valid;
        }
        tail_rows(pass_k(P.t_step0), true);
    }
    __syncthreads();
    SDFR_STAMP(8, 1);
    const bool jdir = MLDS && __builtin_amdgcn_readfirstlane((int)(P.L[0].in_dim <= 8 && NI <= 8)) != 0;
    if (MLDS && jdir) {
        for (int e = tid; e < PT * 8; e += NT) jinj[e] = 0.f;
    } else if (JAC) {
        // J rows of this tile start at zero: every contribution to them (layer-0 in-gradient, re-injected columns) is an atomicAdd issued by
        // this workgroup after later barriers, so no separate memset launch is needed
        for (int e = tid; e < PT * NI; e += NT) {
            const int pt = e / NI;
            if (slots[pt] >= 0) P.J[(int64_t)slots[pt] * NI + (e - pt * NI)] = 0.f;
        }
    }

    auto build_operand = [&]() {
        // ---- layer-0 operand: act[k][pt] = inputs[row(pt)][k], zero padded to the K tile ---------------------
        if (!GMASK) {
            const int k0pad = HALF ? P.L[0].kp_h : P.L[0].kp_f;
            for (int e = tid; e < PT * k0pad; e += NT) {
                const int pt = e / k0pad, k = e - pt * k0pad;
                const float v = (k < NI) ? P.inputs[(int64_t)rows[pt] * NI + k] : 0.f;
                act_e[((k / KV) * PT + pt) * KV + (k % KV)] = (ET)v;

            }
        }
        if (KGX > KG) {
            for (int e = tid; e < PT * 4 * KV; e += NT) {
                const int pt = e / (4 * KV), k = e - pt * (4 * KV);
                const float v = (P.kinj && k < NI) ? P.inputs[(int64_t)rows[pt] * NI + k] : 0.f;
                act_e[((KG + k / KV) * PT + pt) * KV + (k % KV)] = (ET)v;
                if (DBUF) actw_e[((KG + k / KV) * PT + pt) * KV + (k % KV)] = (ET)v;       // (the tile's input rows sit behind BOTH operand tiles)
            }
        }
    };
    build_operand();
    __syncthreads();

    const int fbase = wave * MS * FT;      // first feature row owned by this wave
    if constexpr (MLDS) {
        // This wave's 64 features of a row and layer are dwords 2 wave, 2 wave + 1 of the row's 64 bytes (layout v2): ONE 8-byte load per (row,
        // layer).  The four lanes that share a row (lane groups lg = 0..3) divide the layers among themselves -- lane group lg fetches layers
        // lg, lg + 4, ... -- and park the words in LDS, [wave][layer][row] (wave-private: producer and consumer lanes sit in the same wave, whose
        // LDS operations execute in order; the barriers in front of the first use are there anyway).  2 NP loads per lane for 8 layers.
        uint2* mst = reinterpret_cast<uint2*>(mlds) + (int64_t)wave * SDFR_MAX_LAYERS * PT;
        const uint2* mb = reinterpret_cast<const uint2*>(P.maskbuf);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t r = rows[p * MS + lp];
            for (int ml = lg; ml < P.n_mfma; ml += NLG)
                mst[ml * PT + p * MS + lp] = mb[(sdfr_mask_dword(r, ml, P.n_mfma, HP / 32, fbase)) >> 1];
        }
    }
    acc_t acc[FT][NP];
    // MLDS kernels: the first weight fragments of the NEXT product are requested before the epilogue of the current one (they depend on
    // nothing), so a layer's first matrix instruction does not wait for an L2 round trip (~1 k cycles of an idle matrix pipe per layer)
    constexpr bool APRE = MLDS;
    vec_t apre[APRE ? FT : 1];
    bool have_pre = false;
    auto preload_a = [&](const vec_t* __restrict__ Wl) {
        if constexpr (APRE) {
            const vec_t* ap = Wl + lg * HP + fbase + lp;
#pragma unroll
            for (int f = 0; f < FT; ++f) apre[f] = ap[f * MS];
            have_pre = true;
        }
    };

    // One transposed GEMM over a K extent of `kpad` (a multiple of KT): acc[f][p] += W_tile(rows fbase+f*MS..) x act.
    // FULL: all FT feature tiles of this wave are active (straight-line MFMA stream, no branches);
    // otherwise only the first `nact` tiles are (thin layers: the 6-wide first layer's backward, small nets).
    auto gemm_body = [&](const vec_t* __restrict__ Wl, int kpad, int nact, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        // activation fragments come from LDS (short latency): their ring PFB may be shallower than the weights' (PF must be a multiple)
        constexpr int PFB = PFB_ > 0 ? PFB_ : ((PF % 2 == 0) ? 2 : PF);
        static_assert(PF % PFB == 0, "PF must be a multiple of PFB");
        const int nkt = kpad / KT;
        const vec_t* aptr = Wl + lg * HP + fbase + lp;
        const vec_t* bptr = act + lg * PT + lp;
        vec_t a[PF][FT], b[PFB][NP];
        auto load_a = [&](int tile, vec_t* aa) {
#ifdef SDFR_PIN_WEIGHTS
            tile &= 1;            // ablation (timing only, wrong results): the weight stream hits L1 -- what the product loop costs without its L2 traffic
#endif
            const vec_t* ap = aptr + (int64_t)tile * (NLG * HP);
#pragma unroll
            for (int f = 0; f < FT; ++f)
                if (FULL || f < nact) aa[f] = ap[f * MS];
        };
        auto load_b = [&](int tile, vec_t* bb) {
            const vec_t* bp = bptr + tile * (NLG * PT);
#pragma unroll
            for (int p = 0; p < NP; ++p) bb[p] = bp[p * MS];
        };
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int i = 0; i < KV; ++i) a[u][f][i] = (ET)0;
        bool took_pre = false;
        if constexpr (APRE && FULL) {
            if (have_pre) {
#pragma unroll
                for (int f = 0; f < FT; ++f) a[0][f] = apre[f];
                took_pre = true;
            }
            have_pre = false;
        }
#pragma unroll
        for (int u = 0; u < PF - 1; ++u)
            if (u < nkt && !(u == 0 && took_pre)) load_a(u, a[u]);
#pragma unroll
        for (int u = 0; u < PFB - 1; ++u)
            if (u < nkt) load_b(u, b[u]);
#ifndef SDFR_STRAIGHT_KLOOP
#define SDFR_STRAIGHT_KLOOP 1
#endif
#ifndef SDFR_UNROLLED_KLOOP
#define SDFR_UNROLLED_KLOOP 0
#endif
        constexpr int NKT_FULL = HP / KT;                 // K tiles of a full-width layer (32 with half operands, 64 with float32)
        if (SDFR_UNROLLED_KLOOP && HALF && FULL && nkt == NKT_FULL && NKT_FULL % PF == 0) {
            // The same loop with a compile-time trip count, fully unrolled: in straight-line code the compiler's s_waitcnt counts are exact for
            // EVERY stage.  In the rolled loop it merges the wait states over the back edge and drains the weight ring to its newest stage at
            // the top of every iteration (vmcnt(2) where vmcnt(6) was due with a ring of 4: profiles/r04_notes.md section 10) -- which is why
            // deeper weight rings never paid before.
#pragma unroll
            for (int t = 0; t < NKT_FULL; t += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    if (t + u + PF - 1 < NKT_FULL) load_a(t + u + PF - 1, a[(u + PF - 1) % PF]);
                    if (t + u + PFB - 1 < NKT_FULL) load_b(t + u + PFB - 1, b[(u + PFB - 1) % PFB]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < M::NSTEP; ++ks)
#pragma unroll
                        for (int f = 0; f < FT; ++f)
#pragma unroll
                            for (int p = 0; p < NP; ++p) acc[f][p] = M::step(a[u][f], b[u % PFB][p], acc[f][p], ks);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
        if (SDFR_STRAIGHT_KLOOP && FULL && nkt % PF == 0) {
            // The common case (every 512-wide layer): a K loop WITHOUT branches.  Prefetch indices are clamped instead of guarded (the
            // tail re-reads the last tile into ring slots nobody consumes), so the body is one basic block and the compiler's s_waitcnt
            // counts stay exact: with the guarded form below it falls back to vmcnt(0) once per PF tiles -- a wait for loads it has just
            // issued, i.e. a full L2 round trip exposed per ring turn (forward on the grid: 1.80 -> 1.71 ms with the branch-free body).
            const int last = nkt - 1;
            for (int t = 0; t < nkt; t += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    load_a(min(t + u + PF - 1, last), a[(u + PF - 1) % PF]);
                    load_b(min(t + u + PFB - 1, last), b[(u + PFB - 1) % PFB]);
                    __builtin_amdgcn_sched_barrier(0);      // keep each stage's prefetch ahead of its products (the scheduler otherwise
                                                            // gathers all loads of the unrolled body in one place and waits on them at once)
#pragma unroll
                    for (int ks = 0; ks < M::NSTEP; ++ks)
#pragma unroll
                        for (int f = 0; f < FT; ++f)
#pragma unroll
                            for (int p = 0; p < NP; ++p) acc[f][p] = M::step(a[u][f], b[u % PFB][p], acc[f][p], ks);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
        for (int t = 0; t < nkt; t += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (t + u < nkt) {
                    if (t + u + PF - 1 < nkt) load_a(t + u + PF - 1, a[(u + PF - 1) % PF]);
                    if (t + u + PFB - 1 < nkt) load_b(t + u + PFB - 1, b[(u + PFB - 1) % PFB]);
#pragma unroll
                    for (int ks = 0; ks < M::NSTEP; ++ks)
#pragma unroll
                        for (int f = 0; f < FT; ++f)
                            if (FULL || f < nact) {
#pragma unroll
                                for (int p = 0; p < NP; ++p) acc[f][p] = M::step(a[u][f], b[u % PFB][p], acc[f][p], ks);
                            }
                }
            }
        }
    };
    // K order of the 16x16x4 kernels, reproduced with 32x32x2 instructions.  A 16x16x4 MFMA sums the four k values of its four lane groups in
    // lane-group order, so a 16-row kernel accumulates k = 16t + 4g + s in the order (t, s, g).  A 32x32x2 instruction holds two lane groups
    // (fragment tile u: k = 8u + 4g + s); walking fragment tiles in PAIRS with the component s as the outer loop -- (u = 2t, s), (u = 2t+1, s)
    // -- visits the same k sequence, so the 32-row Jacobian kernel (many crops per launch) returns bit for bit what the 16-row one (few crops)
    // does, and results stay independent of the batch size.  Ring of two pairs; branch-free when the pair count is even.
    auto gemm_pair = [&](const vec_t* __restrict__ Wl, int kpad, int nact, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int npair = kpad / (2 * KT);
        const vec_t* aptr = Wl + lg * HP + fbase + lp;
        const vec_t* bptr = act + lg * PT + lp;
        vec_t a[4][FT], b[4][NP];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int i = 0; i < KV; ++i) a[u][f][i] = (ET)0;
        auto load_pair = [&](int pr, int slot) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const vec_t* ap = aptr + (int64_t)(2 * pr + h) * (NLG * HP);
                const vec_t* bp = bptr + (2 * pr + h) * (NLG * PT);
#pragma unroll
                for (int f = 0; f < FT; ++f)
                    if (FULL || f < nact) a[slot + h][f] = ap[f * MS];
#pragma unroll
                for (int p = 0; p < NP; ++p) b[slot + h][p] = bp[p * MS];
            }
        };
        auto compute = [&](int slot) {
#pragma unroll
            for (int ks = 0; ks < M::NSTEP; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h)          // per accumulator the order stays (s, first tile), (s, second tile); the accumulators alternate
#pragma unroll
                    for (int f = 0; f < FT; ++f)
                        if (FULL || f < nact) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) acc[f][p] = M::step(a[slot + h][f], b[slot + h][p], acc[f][p], ks);
                        }
        };
        if (npair <= 0) return;
        load_pair(0, 0);
        if (FULL && (npair % 2 == 0)) {
            const int lastp = npair - 1;
            for (int pr = 0; pr < npair; pr += 2) {
                load_pair(pr + 1, 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(0);
                __builtin_amdgcn_sched_barrier(0);
                load_pair(min(pr + 2, lastp), 0);
                __builtin_amdgcn_sched_barrier(0);
                compute(2);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        for (int pr = 0; pr < npair; ++pr) {           // thin or odd layers: plain loop, same order
            if (pr > 0) load_pair(pr, 0);
            compute(0);
        }
    };
    constexpr bool KPAIR = (MS == 32) && (MODE == 3) && !HALF && (HP == 512);
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
        if constexpr (KPAIR) {
            if (nact == FT) gemm_pair(Wl, kpad, FT, std::true_type{});
            else if (nact > 0) gemm_pair(Wl, kpad, nact, std::false_type{});
        } else {
            if (nact == FT) gemm_body(Wl, kpad, FT, std::true_type{});
            else if (nact > 0) gemm_body(Wl, kpad, nact, std::false_type{});
        }
    };
    // first feature of the 4-register group rg of feature tile f held by this lane
    auto feat0 = [&](int f, int rg) { return fbase + f * MS + rg * (4 * NLG) + 4 * lg; };
    const vec_t* Wfwd = reinterpret_cast<const vec_t*>(HALF ? (const void*)P.Wh : (const void*)P.Wf);

    // ---- LayerNorm helpers (LN decoders only) ------------------------------------------------------------------
    // sum over ALL features of a per-thread partial, per point of the tile: lane groups by shuffle, waves through LDS
    auto point_sums = [&](float* v) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float x = v[p];
            for (int o = MS; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
            if (lg == 0) lnred[wave * PT + p * MS + lp] = x;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += lnred[w * PT + p * MS + lp];
            v[p] = t;
        }
        __syncthreads();
    };
    const int64_t wg_lin = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    // acc (linear output without bias) -> gamma * x_hat + beta ; JAC: x_hat and rstd are kept for the backward
    auto ln_forward = [&](int l, int n_out) {
        const float* bias = P.bias + l * HP;
        const float* gam = P.ln_gamma + l * HP;
        const float* bet = P.ln_beta + l * HP;
        float s1[NP], mu[NP], rs[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) s1[p] = 0.f;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int r = 0; r < RG * 4; ++r) {
                const int j = feat0(f, r >> 2) + (r & 3);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float x = (j < n_out) ? acc[f][p][r] + bias[j] : 0.f;
                    acc[f][p][r] = x;
                    s1[p] += x;
                }
            }
        point_sums(s1);
#pragma unroll
        for (int p = 0; p < NP; ++p) { mu[p] = s1[p] / (float)n_out; s1[p] = 0.f; }
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int r = 0; r < RG * 4; ++r) {
                const int j = feat0(f, r >> 2) + (r & 3);
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    if (j < n_out) { const float d = acc[f][p][r] - mu[p]; s1[p] += d * d; }
            }
        point_sums(s1);
#pragma unroll
        for (int p = 0; p < NP; ++p) rs[p] = 1.f / sqrtf(s1[p] / (float)n_out + 1e-5f);
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int j0 = feat0(f, rg);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    float xh[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = j0 + i;
                        xh[i] = (j < n_out) ? (acc[f][p][rg * 4 + i] - mu[p]) * rs[p] : 0.f;
                        acc[f][p][rg * 4 + i] = (j < n_out) ? xh[i] * gam[j] + bet[j] : 0.f;
                    }
                    if (JAC) P.ln_ws[((wg_lin * P.n_mfma + l) * (HP / 4) + (j0 >> 2)) * PT + p * MS + lp] = make_float4(xh[0], xh[1], xh[2], xh[3]);
                }
            }
        if (JAC && lg == 0 && wave == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) lnrstd[l * PT + p * MS + lp] = rs[p];
        }
    };

    // ---- forward through the MFMA layers -------------------------------------------------------------------
    int* t_more = reinterpret_cast<int*>(gy);           // gy is unused by forward modes
    do {
    for (int l = 0; !GMASK && l < P.n_mfma; ++l) {
        const MlpLayer L = P.L[l];
        const MlpLayer Ln = P.L[l + 1];
        SDFR_STAMP(l, 0);
        gemm(Wfwd + (HALF ? L.off_h : L.off_f), HALF ? L.kp_h : L.kp_f, L.out_dim);
        SDFR_STAMP(l, 1);
        // all bias vectors of this lane are requested before the barrier, so their L2 latency overlaps the wait for the other waves
        // (left inside the store loop the compiler serialises them: one exposed round trip per register group)
        const float* bias = P.bias + l * HP;
        float4 b4s[FT][RG];
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) b4s[f][rg] = *reinterpret_cast<const float4*>(bias + feat0(f, rg));
        if (!DBUF) __syncthreads();                       // every wave is done reading act (two operand tiles: the epilogue writes the other one)
        SDFR_STAMP(l, 2);
        uint32_t mw[MW];
#pragma unroll
        for (int w = 0; w < MW; ++w) mw[w] = 0u;
        const int inj_lo = L.out_dim, inj_hi = L.out_dim + Ln.inj_n;
        bool lnl = false;
        if constexpr (LN) {
            lnl = L.ln != 0;
            if (lnl) ln_forward(l, L.out_dim);          // acc already holds gamma * x_hat + beta (bias included)
        }
        // Re-injected input columns (latent_in / xyz_in_all) occupy a few features of ONE wave in ONE or few layers: whether this wave's
        // feature block touches them is a scalar question, asked once, so that every other wave and layer runs an epilogue without the
        // per-group lane-divergent range checks (32 saveexec/branch pairs per layer otherwise).
        const bool inj_here = __builtin_amdgcn_readfirstlane((int)((Ln.inj_n > 0) && !(KGX > KG && P.kinj) && (fbase + MS * FT > inj_lo) &&
                                                                    (fbase < inj_hi))) != 0;
        auto epilogue = [&](auto inj_tag) {
            constexpr bool INJ = decltype(inj_tag)::value;
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    const int j0 = feat0(f, rg);
                    const float4 b4 = lnl ? make_float4(0.f, 0.f, 0.f, 0.f) : b4s[f][rg];
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
                        if (INJ) {
                            if (j0 + 3 >= inj_lo && j0 < inj_hi) {    // re-inject input columns for the next layer
                                const float* src = P.inputs + (int64_t)rows[pt] * NI + Ln.inj_off - inj_lo;
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (j0 + i >= inj_lo && j0 + i < inj_hi) v[i] = src[j0 + i];
                            }
                        }
                        store4(actw_e + ((j0 / KV) * PT + pt) * KV + (j0 % KV), v);
                    }
                }
        };
        // Half-operand forward, plain layers (no re-injection in this wave, no LayerNorm): the epilogue is VALU-issue bound (two waves per
        // SIMD, ~7 instructions per value in the generic form above against 16 k cycles of matrix work per layer), so it is written for
        // instruction count: bias add, ONE v_alignbit per value shifting its sign bit into the mask word (bit = "not negative"; the words are
        // bit-reversed and inverted once per 32 values), packed f32->f16 conversion and a packed half max for the ReLU (max(rne(x), 0) =
        // rne(max(x, 0))).  Loop order f, p, rg, i = ascending mask bit.
        auto epilogue_half_fast = [&]() {
            uint32_t run = 0u;
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int pt = p * MS + lp;
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg) {
                        const int j0 = feat0(f, rg);
                        const float4 b4 = b4s[f][rg];
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef h16 h16x2 __attribute__((ext_vector_type(2)));
                        float x[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            x[i] = acc[f][p][rg * 4 + i] + f4c(b4, i);
                            if (LMASK || SAVE) {
                                run = __builtin_amdgcn_alignbit(run, __float_as_uint(x[i]), 31);      // run = run << 1 | sign(x)
                                const int bit = ((f * NP + p) * RG + rg) * 4 + i;
                                if ((bit & 31) == 31) mw[bit >> 5] = ~__builtin_bitreverse32(run);
                            }
                        }
                        f32x2 lo2 = {x[0], x[1]}, hi2 = {x[2], x[3]};
                        h16x2 lo = __builtin_convertvector(lo2, h16x2), hi = __builtin_convertvector(hi2, h16x2);
                        const h16x2 z = {(h16)0, (h16)0};
                        lo = __builtin_elementwise_max(lo, z);
                        hi = __builtin_elementwise_max(hi, z);
                        h16x4 t;
                        t[0] = lo[0]; t[1] = lo[1]; t[2] = hi[0]; t[3] = hi[1];
                        *reinterpret_cast<h16x4*>(actw_e + ((j0 / KV) * PT + pt) * KV + (j0 % KV)) = t;
                    }
#ifdef SDFR_EPI_FENCE
                    __builtin_amdgcn_sched_barrier(0);         // (A/B: one accumulator tile's reads at a time -- 512-register geometries)
#endif
                }
        };
#ifndef SDFR_H_FAST_EPI
#define SDFR_H_FAST_EPI 1
#endif
        if constexpr (SDFR_H_FAST_EPI && HALF && !LN && !JAC) {
#ifdef SDFR_ABL_NOEPI
            if (acc[0][0][0] == 12345.f) epilogue_half_fast();  // ablation (timing only, wrong results): barriers stay, the epilogue's work goes
#else
            if (inj_here) epilogue(std::true_type{});          // decoders with more than 8 input columns: generic injection
            else epilogue_half_fast();
#endif
        }
        else if (inj_here) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (LMASK) {
#pragma unroll
            for (int w = 0; w < MW; ++w) masks[(l * MW + w) * NT + tid] = mw[w];
        }
        if (SAVE && P.maskbuf) {
            // layout v2 (sdfr_mask_dword): this thread's 16 bits of (feature tile f, point tile p) -- bits ((f*NP + p)*4 + rg)*4 + i of its mask
            // words -- are one short of row tile*PT + p*32 + lp, dword (fbase >> 5) + f, half lg.  The 128-row block is uniform over the workgroup.
            const int64_t row0 = (int64_t)tile * PT;
            if constexpr (FT == 2) {
                // The wave's 64 features are dwords 2 wave, 2 wave + 1 of a row's line; lanes lp and lp + 32 (lg = 0 / 1) hold the two 16-bit halves
                // of both.  One cross-lane exchange per point tile gives BOTH lanes the row's full 8 bytes, and lane group p & 1 stores them: NP / 2
                // 8-byte stores per lane and layer, 32 lanes x 8 bytes per instruction.  (r06's first layout-v2 builds wrote FT NP 16-bit stores per
                // lane -- 32 partial writes of 2 bytes per 64-byte line and layer: the half forward with masks lost 11-16 %, 93 -> 108 us at one crop.)
                uint2* mb2 = reinterpret_cast<uint2*>(P.maskbuf) + (((row0 >> 7) * P.n_mfma + l) * 128) * (int64_t)(HP / 64);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const uint32_t h0 = (mw[p >> 1] >> ((p & 1) * 16)) & 0xffffu;                    // f = 0: field f * NP + p
                    const uint32_t h1 = (mw[(NP + p) >> 1] >> (((NP + p) & 1) * 16)) & 0xffffu;      // f = 1
                    const uint32_t mine = h0 | (h1 << 16);
                    const uint32_t other = (uint32_t)__shfl_xor((int)mine, 32, 64);
                    const uint32_t lo = lg == 0 ? mine : other, hi = lg == 0 ? other : mine;         // lane group 0's / 1's halves
                    if (lg == (p & 1))
                        mb2[((int)(row0 & 127) + p * MS + lp) * (HP / 64) + (fbase >> 6)] = make_uint2((lo & 0xffffu) | (hi << 16), (lo >> 16) | (hi & 0xffff0000u));
                }
            } else {
            uint16_t* mb = reinterpret_cast<uint16_t*>(P.maskbuf) + (((row0 >> 7) * P.n_mfma + l) * 128) * (int64_t)(HP / 32) * 2;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int off = (((int)(row0 & 127) + p * MS + lp) * (HP / 32) + (fbase >> 5)) * 2 + lg;
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const int fi = f * NP + p;
                    mb[off + 2 * f] = (uint16_t)(mw[fi >> 1] >> ((fi & 1) * 16));
                }
            }
            }
        }
        SDFR_STAMP(l, 3);
        __syncthreads();
        if (DBUF) { vec_t* t_ = act; act = actw; actw = t_; ET* e_ = act_e; act_e = actw_e; actw_e = e_; }       // the next layer reads what this one wrote
        SDFR_STAMP(l, 4);
    }

    // ---- last linear (H -> 1) + tanh -----------------------------------------------------------------------
    if (GMASK) {
        if (tid < PT) {
            const float o = P.sdf_in[rows[tid]];
            gy[tid] = 1.f - o * o;
            if (tid < n_valid && P.sdf_sel) P.sdf_sel[slots[tid]] = o;
        }
    } else {
        // k slices of the last linear's dot product.  Half operands: ONE partition (4 slices of 128 k, summed in order) for every tile
        // geometry, so that a row's value has the same bits whether a 128-, 64- or 16-row tile evaluated it (r04: the sphere tracer's
        // march picks the tile size by the device-side row count, and a crop must march the same alone and in a batch); float32: as many
        // slices as the workgroup has threads per point (unchanged bits).
        constexpr int SL = HALF ? (NT / PT < 4 ? NT / PT : 4) : NT / PT;       // (fewer than 4 threads per point: A/B geometries only)
        static_assert(SL * PT <= NT, "last linear: one thread per (slice, point)");
        constexpr int KGS = KG / SL;
        const int sl = tid / PT, pt = tid - sl * PT;
        float s = 0.f;
        if (sl < SL) {
            const float* wl = P.w_last + sl * KGS * KV;
            const vec_t* a4 = act + (sl * KGS) * PT + pt;
#pragma unroll 4
            for (int g = 0; g < KGS; ++g) {
                const vec_t a = a4[g * PT];
#pragma unroll
                for (int i = 0; i < KV; ++i) s = fmaf((float)a[i], wl[g * KV + i], s);
            }
        }
        red[tid] = s;
        __syncthreads();
        if (tid < PT) {
            float y = 0.f;
#pragma unroll
            for (int q = 0; q < SL; ++q) y += red[q * PT + tid];
            if (KGX > KG && P.kinj) {
                // K-side injection: input columns concatenated in front of the LAST linear (xyz_in_all) are not in the feature slots; their
                // products come from the tile's input rows kept behind them
                const MlpLayer Ll = P.L[P.n_mfma];
                for (int c = 0; c < Ll.inj_n; ++c) {
                    const int kc = Ll.inj_off + c;
                    y = fmaf((float)act_e[((KG + kc / KV) * PT + tid) * KV + (kc % KV)], P.w_last[P.L[P.n_mfma - 1].out_dim + c], y);
                }
            }
            y += P.b_last;
            const float y1 = P.use_tanh ? tanhf(y) : y;
            const float o = tanhf(y1);
            if (JAC) {
                float g = 1.f - o * o;
                if (P.use_tanh) g *= (1.f - y1 * y1);
                gy[tid] = g;
                if (tid < n_valid && P.sdf_sel) P.sdf_sel[slots[tid]] = o;
            } else if (TAIL) {
                // The step rule of the march (oracle/sdf_oracle.py::sphere_trace; K = 1: exactly sdfr_trace_step_kernel's trace_advance).  Row
                // j*RT + i carries sample j of ray i: lane i < RT collects its K values, walks the accepted prefix -- sample j counts only
                // inside the safe sphere of sample j-1 --, retires hits and exits, continues from the last accepted sample.
                const int k = pass_k(P.t_step0 + t_pass);
                const unsigned long long live = __ballot(t_act && tid < RT);
                if (tid == 0) t_ev += (unsigned long long)(__popcll(live) * k);
                {
                    const int li = tid < RT ? tid : 0;
                    float vj = o, rj = fabsf(o), pj = t_st.x, prev = t_st.y;
                    bool done = rj < P.t_eps, ishit = done;
                    float pc = t_st.x, qp = t_st.z, vp = o;                 // position / ratio power / value of sample j-1
                    for (int j = 1; j < k; ++j) {                           // (uniform trip count: the shuffles run in every lane)
                        const float vc = __shfl(o, li + RT * j, 64);
                        const float pn = pc + ((P.t_sigma * qp) * t_st.y) / t_idn;
                        qp = qp * t_st.z;
                        const float rp = fabsf(vp);
                        const bool cov = ((pn - pc) * t_idn <= rp) && (vp > 0.f) && (pn > pc);
                        const bool take = cov && !done;
                        done = done || !cov;
                        if (take) {
                            prev = rp; vj = vc; rj = fabsf(vc); pj = pn;
                            if (rj < P.t_eps) { ishit = true; done = true; }
                        }
                        pc = pn; vp = vc;
                    }
                  if (t_act && tid < RT) {
                    if (ishit) {
                        P.t_hit_lam[t_gp] = pj;
                        P.t_hit_sdf[t_gp] = vj;
                        t_act = false;
                    } else {
                        const float qn = (prev > 0.f) ? fminf(fmaxf(rj / prev, 0.5f), P.t_qmax) : 1.f;
                        const float l2 = pj + vj / t_idn;
                        t_st.y = rj; t_st.z = qn;
                        if ((l2 < t_farl) && (vj == vj)) t_st.x = l2;
                        else t_act = false;
                    }
                  }
                }
            } else {
                if (tid < n_valid) {
                    const int64_t row = (int64_t)tile * PT + tid;
                    if (!P.skip || !P.skip[row / P.skip_rows]) P.sdf[row] = o;      // a flagged crop keeps its (exact, patched) values
                }
            }
        }
    }
    if constexpr (TAIL) {
        // another pass while a ray of the tile is still marching and the step budget lasts
        --t_left;
        if (tid == 0) *t_more = 0;
        __syncthreads();
        if (t_act && t_left > 0) *t_more = 1;
        __syncthreads();
        if (*t_more == 0) break;
        ++t_pass;
        tail_rows(pass_k(P.t_step0 + t_pass), false);
        __syncthreads();
        build_operand();                                  // the rows this workgroup has just rewritten (same CU: its L1 is coherent for it)
        __syncthreads();
    }
    } while (TAIL);
    if constexpr (TAIL) {
        // rays still marching: out of budget -> unresolved; end of the stage -> appended to the next stage's list (the order of the tiles'
        // appends varies from run to run -- which rays share a tile does not enter any ray's arithmetic)
        if (tid < 64) {
            const unsigned long long left_over = __ballot(t_act && tid < RT);
            const bool hand_over = P.t_next_cnt && (P.t_stage < P.t_steps);
            int base = 0;
            if (tid == 0) {
                if (P.t_evals) atomicAdd(P.t_evals, t_ev);
                if (left_over && !hand_over && P.t_unresolved) atomicAdd(P.t_unresolved, (int)__popcll(left_over));
                if (left_over && hand_over) base = atomicAdd(P.t_next_cnt, (int)__popcll(left_over));
            }
            if (hand_over && left_over) {
                base = __shfl(base, 0, 64);
                if (t_act && tid < RT) {
                    const int at = base + (int)__popcll(left_over & ((1ull << tid) - 1ull));
                    P.t_next_pix[at] = t_gp;
                    P.t_next_lam[at] = t_st;
                }
            }
        }
        if (!P.t_tile_ctr) return;                         // (one tile per workgroup: launches without a tile counter)
        continue;                                          // the pool's next tile
    }
    if constexpr (JAC) {
    __syncthreads();
    SDFR_STAMP(8, 2);

    // ---- backward: d out / d inputs for every point of the tile ---------------------------------------------
    // in-gradient of layer l (features k = in-features of layer l) -> masked operand for layer l-1, or J
    // MODE 3 (kernels without MLDS): the saved mask dword of layer l-1 for this lane's features and point (layout v2: sdfr_mask_dword), fetched
    // from HBM/L2.  Issued BEFORE the product of layer l (they do not depend on it), so their latency is hidden behind the K loop instead of
    // sitting in front of the epilogue.
    auto mask_addr = [&](int j, int r, int l, int& shift) -> int64_t {
        shift = sdfr_mask_shift(j);
        return sdfr_mask_dword(r, l - 1, P.n_mfma, HP / 32, j);
    };
    auto fetch_masks = [&](int l, uint32_t* raw) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = rows[p * MS + lp];
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    int shift;
                    raw[(p * FT + f) * RG + rg] = P.maskbuf[mask_addr(feat0(f, rg), r, l, shift)];
                }
        }
    };
    auto store_in_grad = [&](int l, const uint32_t* raw, auto&& value) {
        const MlpLayer L = P.L[l];
        const int prev_out = P.L[l - 1].out_dim;
        const int inj_hi = prev_out + L.inj_n;
        if constexpr (MLDS) {
            // masks of layer l-1 from LDS: the row's two dwords of this wave's features; feature tile f, value i of this lane is bit
            // sdfr_mask_shift(16 f + 4 lg + i) = (lg & 1) * 16 + (lg >> 1) * 4  +  (f & 1) * 8 + i  of dword f >> 1 -- one per-lane shift per dword,
            // then compile-time positions: a value costs one v_bfe_i32 (0 / -1) and one v_and
            const uint2* mst = reinterpret_cast<const uint2*>(mlds) + ((int64_t)wave * SDFR_MAX_LAYERS + (l - 1)) * PT;
            const int msh = (lg & 1) * 16 + (lg >> 1) * 4;
            uint32_t mk[NP][2];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint2 m2 = mst[p * MS + lp];
                mk[p][0] = m2.x >> msh; mk[p][1] = m2.y >> msh;
            }
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                const int j0 = feat0(f, 0);
                // features >= prev_out (padding, re-injected input columns) sit in the last feature tile(s) of the last wave of one or two layers
                const bool tail_f = __builtin_amdgcn_readfirstlane((int)(fbase + (f + 1) * MS > prev_out)) != 0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int pt = p * MS + lp;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = value(f, p, 0, i, j0 + i, pt);
                        const int m = __builtin_amdgcn_sbfe((int)mk[p][f >> 1], (f & 1) * 8 + i, 1);
                        v[i] = __int_as_float(__float_as_int(x) & m);
                    }
                    if (tail_f) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = j0 + i;
                            if (k >= prev_out) {
                                if (k < inj_hi) {
                                    // (one lane per (point, column) and layer; layers are separated by barriers: a plain LDS update)
                                    if (jdir) jinj[pt * 8 + L.inj_off + (k - prev_out)] += value(f, p, 0, i, k, pt);
                                    else if (slots[pt] >= 0) atomicAdd(P.J + (int64_t)slots[pt] * NI + L.inj_off + (k - prev_out), value(f, p, 0, i, k, pt));
                                }
                                v[i] = 0.f;
                            }
                        }
                    }
                    store4(act_e + ((j0 / KV) * PT + pt) * KV + (j0 % KV), v);
                }
            }
            return;
        }
        uint32_t mw[MW];
        if (GMASK) {
#pragma unroll
            for (int w = 0; w < MW; ++w) mw[w] = 0u;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int r = rows[p * MS + lp];
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg) {
                        int shift;
                        (void)mask_addr(feat0(f, rg), r, l, shift);
                        const uint32_t nib = (raw[(p * FT + f) * RG + rg] >> shift) & 0xFu;
                        const int bit = ((f * NP + p) * RG + rg) * 4;
                        mw[bit >> 5] |= nib << (bit & 31);
                    }
            }
        } else {
#pragma unroll
            for (int w = 0; w < MW; ++w) mw[w] = masks[((l - 1) * MW + w) * NT + tid];
        }
        // 1) ReLU mask of layer l-1 (its output features), re-injected input columns -> J.  Features >= prev_out (padding and the
        // re-injected input columns of latent_in / xyz_in_all layers) sit in the last wave's block of one or two layers: whether this wave's
        // block reaches them is a scalar question asked once, so every other wave and layer runs the plain masking without per-element range
        // checks and lane-divergent atomics.
        const bool tail_here = __builtin_amdgcn_readfirstlane((int)(fbase + MS * FT > prev_out)) != 0;
        auto apply_masks = [&](auto tail_tag) {
            constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    const int j0 = feat0(f, rg);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int pt = p * MS + lp;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = j0 + i;
                            const int bit = ((f * NP + p) * RG + rg) * 4 + i;
                            float x = value(f, p, rg, i, k, pt);
                            if (!TAIL || k < prev_out) {
                                x = ((mw[bit >> 5] >> (bit & 31)) & 1u) ? x : 0.f;
                            } else {
                                if (k < inj_hi && slots[pt] >= 0)
                                    atomicAdd(P.J + (int64_t)slots[pt] * NI + L.inj_off + (k - prev_out), x);
                                x = 0.f;
                            }
                            acc[f][p][rg * 4 + i] = x;
                        }
                    }
                }
        };
        if (tail_here) apply_masks(std::true_type{});
        else apply_masks(std::false_type{});
        // 2) LayerNorm backward of layer l-1:  g_x = rstd * (g*gamma - mean(g*gamma) - x_hat * mean(g*gamma*x_hat))
        if constexpr (LN) {
            if (P.L[l - 1].ln) {
                const float* gam = P.ln_gamma + (l - 1) * HP;
                const float4* xh4 = P.ln_ws + ((wg_lin * P.n_mfma + (l - 1)) * (HP / 4)) * PT;
                float s1[NP], s2[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) { s1[p] = 0.f; s2[p] = 0.f; }
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg) {
                        const int j0 = feat0(f, rg);
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const float4 xh = xh4[(j0 >> 2) * PT + p * MS + lp];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int k = j0 + i;
                                const float gh = (k < prev_out) ? acc[f][p][rg * 4 + i] * gam[k] : 0.f;
                                acc[f][p][rg * 4 + i] = gh;
                                s1[p] += gh;
                                s2[p] += gh * f4c(xh, i);
                            }
                        }
                    }
                point_sums(s1);
                point_sums(s2);
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg) {
                        const int j0 = feat0(f, rg);
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const float4 xh = xh4[(j0 >> 2) * PT + p * MS + lp];
                            const float rs = lnrstd[(l - 1) * PT + p * MS + lp];
                            const float m1 = s1[p] / (float)prev_out, m2 = s2[p] / (float)prev_out;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[f][p][rg * 4 + i] = (j0 + i < prev_out) ? rs * (acc[f][p][rg * 4 + i] - m1 - f4c(xh, i) * m2) : 0.f;
                        }
                    }
            }
        }
        // 3) operand of the next (previous layer's) transposed product
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int j0 = feat0(f, rg);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[f][p][rg * 4 + i];
                    store4(act_e + ((j0 / KV) * PT + p * MS + lp) * KV + (j0 % KV), v);
                }
            }
    };

    // top: in-gradient of the last linear = w_last[k] * gy[pt]
    uint32_t raw[NP * FT * RG];
    if (GMASK && !MLDS) fetch_masks(P.n_mfma, raw);
    SDFR_STAMP(8, 3);
    if constexpr (MLDS) {
        if (P.n_mfma >= 1 && !(P.n_mfma - 1 == 0 && P.L[0].in_dim <= 8))
            preload_a(reinterpret_cast<const vec_t*>(P.Wbh) + P.L[P.n_mfma - 1].off_bh);
        float4 wl[FT];
        float gyp[NP];
#pragma unroll
        for (int f = 0; f < FT; ++f) wl[f] = *reinterpret_cast<const float4*>(P.w_last + feat0(f, 0));
#pragma unroll
        for (int p = 0; p < NP; ++p) gyp[p] = gy[p * MS + lp];
        store_in_grad(P.n_mfma, raw, [&](int f, int p, int, int i, int, int) { return f4c(wl[f], i) * gyp[p]; });
    } else
    store_in_grad(P.n_mfma, raw, [&](int, int, int, int, int k, int pt) { return P.w_last[k] * gy[pt]; });
    __syncthreads();
    SDFR_STAMP(8, 4);
    for (int l = P.n_mfma - 1; l >= 0; --l) {
        const MlpLayer L = P.L[l];
        if constexpr (MLDS) {
          if (l == 0 && L.in_dim <= 8) {
            // First layer, at most 8 inputs (latent + xyz): the transposed product has ONE 16-row feature tile, i.e. work for one wave, K tile
            // after K tile.  Here the K extent (the layer's 512 out-features) is split over the 8 waves instead -- each wave two K tiles of the
            // half weight image on the matrix pipe, partial sums through LDS, added in wave order -- the same instruction sequence per point
            // on every tile geometry, so a row's J has the same bits whichever launch computed it.  (r05: a VALU product against the float32
            // image, every lane loading the same 16 bytes per feature: 18 k cycles per 64-row tile, paced by the L1 return path.)
            constexpr int TPW = (HP / KT) / NW;
            const vec_t* Wl = reinterpret_cast<const vec_t*>(P.Wbh) + L.off_bh;
            const int nkt = L.kp_bh / KT;
            acc_t a0[NP];
            vec_t av[TPW], bv[TPW][NP];
            SDFR_STAMP(9, 0);
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const int t = wave * TPW + u;
#pragma unroll
                for (int i = 0; i < KV; ++i) av[u][i] = (ET)0;
                if (t < nkt) av[u] = Wl[(int64_t)(t * NLG + lg) * HP + lp];
#pragma unroll
                for (int p = 0; p < NP; ++p) bv[u][p] = act[(t * NLG + lg) * PT + p * MS + lp];
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[p][r] = 0.f;
#pragma unroll
                for (int u = 0; u < TPW; ++u) a0[p] = M::step(av[u], bv[u][p], a0[p], 0);
            }
            SDFR_STAMP(9, 1);
            // partial J columns 4 lg .. 4 lg + 3 of point p*16 + lp -> [wave][point][8] (the mask words are dead: their LDS is the scratch)
            float* part = reinterpret_cast<float*>(mlds);
            if (lg < 2) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    *reinterpret_cast<float4*>(part + ((wave * PT + p * MS + lp) * 8 + 4 * lg)) = make_float4(a0[p][0], a0[p][1], a0[p][2], a0[p][3]);
            }
            __syncthreads();
            for (int e = tid; e < 8 * PT; e += NT) {
                const int q = e >> 3, k = e & 7;
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += part[(w * PT + q) * 8 + k];
                if (k < NI && slots[q] >= 0) {
                    if (k >= L.in_dim) t = 0.f;
                    if (jdir) P.J[(int64_t)slots[q] * NI + k] = jinj[q * 8 + k] + t;
                    else atomicAdd(P.J + (int64_t)slots[q] * NI + k, t);
                }
            }
            SDFR_STAMP(9, 2);
            break;
          }
        }
        if (!MLDS && l == 0 && L.in_dim <= 8) {
            // The first layer of a DeepSDF decoder has a handful of inputs (latent + xyz): its transposed product is 8 x HP x PT multiply-adds,
            // which one wave would do alone on the matrix pipe, K tile after K tile (20 k cycles of exposed latency).  Here every thread
            // takes one point and HP / (NT / PT) features on the VALU, and the partial sums are added in a fixed order.
            // Partition of the HP in-gradients of a point into PARTS contiguous blocks summed in order, then the blocks in order: for the
            // 512-wide Jacobian variants the partition is fixed whatever the workgroup shape -- 16 blocks for the float32 kernels (threads per
            // point TPP = 16 or 8), 32 for the half ones (TPP = 32 on 16-row tiles, 8 on 64-row tiles) -- so that the tile geometries of one
            // precision round identically.
            constexpr int TPP = NT / PT;
            constexpr int PMIN = (HP == 512) ? (HALF ? 32 : 16) : 1;
            constexpr int PARTS = (TPP < PMIN) ? PMIN : TPP, PPT = PARTS / TPP, JS = HP / PARTS;
            static_assert(NT % PT == 0 && HP % PARTS == 0 && PARTS % TPP == 0 && PARTS * 8 * PT * 4 <= KG * PT * 16,
                          "first-layer reduction scratch fits the operand tile");
            // (tiles of 64 rows or more: a wave's threads share their block of features, so the weight rows are wave-uniform -- scalar loads
            // through the constant cache instead of 128 vector loads of 16 bytes per thread, which the L1 path paced at 18 k cycles per tile)
            const int pt = tid % PT, t0 = (PT % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / PT) : tid / PT;
            const float4* W0 = P.Wf + L.off_f;                   // forward image of layer 0: W0[(k/4)*HP + j] = W[j][k..k+3]
            float s8[PPT][8];
            SDFR_STAMP(9, 0);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int part = t0 * PPT + q;
#pragma unroll
                for (int k = 0; k < 8; ++k) s8[q][k] = 0.f;
#pragma unroll 8
                for (int jj = 0; jj < JS; ++jj) {                // (8 steps unrolled: 16 weight loads in flight -- unrolled by 4 the loop was a chain of 16 exposed L1 / L2 latencies)
                    const int j = part * JS + jj;
                    const float g = (float)act_e[((j / KV) * PT + pt) * KV + (j % KV)];
                    float4 wa, wb;
                    if constexpr (PT % 64 == 0) {
                        // wave-uniform rows of a read-only image: through the constant address space, i.e. scalar loads (s_load_dwordx4) into
                        // SGPRs -- as vector loads every lane fetched the same 16 bytes, and writing 64 copies of them into the VGPRs paced
                        // the whole step at the L1 return path's 64 B/clk (18 k cycles per tile)
                        typedef const f32x4 __attribute__((address_space(4))) * cf4p;
                        const cf4p W0c = (cf4p)(uintptr_t)W0;
                        const f32x4 va = W0c[j];
                        f32x4 vb = {0.f, 0.f, 0.f, 0.f};
                        if (L.kp_f > 4) vb = W0c[HP + j];
                        wa = make_float4(va[0], va[1], va[2], va[3]);
                        wb = make_float4(vb[0], vb[1], vb[2], vb[3]);
                    } else {
                        wa = W0[j];
                        wb = (L.kp_f > 4) ? W0[HP + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    s8[q][0] = fmaf(wa.x, g, s8[q][0]); s8[q][1] = fmaf(wa.y, g, s8[q][1]); s8[q][2] = fmaf(wa.z, g, s8[q][2]); s8[q][3] = fmaf(wa.w, g, s8[q][3]);
                    s8[q][4] = fmaf(wb.x, g, s8[q][4]); s8[q][5] = fmaf(wb.y, g, s8[q][5]); s8[q][6] = fmaf(wb.z, g, s8[q][6]); s8[q][7] = fmaf(wb.w, g, s8[q][7]);
                }
            }
            SDFR_STAMP(9, 1);
            __syncthreads();                                    // every thread has read its in-gradients: the tile becomes scratch
            float* scr = reinterpret_cast<float*>(lds4);
#pragma unroll
            for (int q = 0; q < PPT; ++q)
#pragma unroll
                for (int k = 0; k < 8; ++k) scr[((t0 * PPT + q) * 8 + k) * PT + pt] = s8[q][k];
            __syncthreads();
            for (int e = tid; e < 8 * PT; e += NT) {           // (one trip while 8 PT <= NT; 128-row tiles of 512 threads: two)
                const int k = e / PT, q = e % PT;
                float t = 0.f;
                for (int r = 0; r < PARTS; ++r) t += scr[(r * 8 + k) * PT + q];
                if (k < NI && k < L.in_dim && slots[q] >= 0) atomicAdd(P.J + (int64_t)slots[q] * NI + k, t);
            }
            SDFR_STAMP(9, 2);
            break;
        }
        SDFR_STAMP(l, 0);
        if (GMASK && !MLDS && l > 0) fetch_masks(l, raw);
        gemm(reinterpret_cast<const vec_t*>(HALF ? (const void*)P.Wbh : (const void*)P.Wb) + (HALF ? L.off_bh : L.off_b), HALF ? L.kp_bh : L.kp_b,
             L.in_dim);
        SDFR_STAMP(l, 1);
        __syncthreads();
        SDFR_STAMP(l, 2);
        if (l > 0) {
            if constexpr (MLDS) {
                if (!(l - 1 == 0 && P.L[0].in_dim <= 8)) preload_a(reinterpret_cast<const vec_t*>(P.Wbh) + P.L[l - 1].off_bh);
            }
            store_in_grad(l, raw, [&](int f, int p, int rg, int i, int, int) { return acc[f][p][rg * 4 + i]; });
            SDFR_STAMP(l, 3);
            __syncthreads();
            SDFR_STAMP(l, 4);
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
    } while (TAIL || PERSIST);
}


// launchers, one translation unit per kernel family (co-compiled instantiations of one template perturb each other's register
// allocation and scheduling by several percent -- CDNA guide, methodology rule 19 -- so the hot kernels are compiled alone)
void sdfr_launch_fwd_f32_512(const MlpParams& P, int64_t n, bool save_masks, hipStream_t s);    // mlp_fwd32.hip
int sdfr_fwd_f32_512_np();                                                                        // its point tiles per workgroup
void sdfr_launch_fwd_f16_512(const MlpParams& P, int64_t n, bool save_masks, hipStream_t s);   // mlp_fwd16.hip
void sdfr_launch_fwd_f16_512_half_tiles(const MlpParams& P, int64_t n, hipStream_t s);          // mlp_fwd16.hip (64-row tiles, masks)
int sdfr_fwd_f16_512_np();                                                                        // its point tiles per workgroup
void sdfr_launch_fwd_split_512(const MlpParams& P, int64_t n, bool save_masks, hipStream_t s);  // mlp_split.hip
void sdfr_launch_jac_f32_512(const MlpParams& P, int cap, int B, bool from_masks, hipStream_t s); // mlp_jac.hip
void sdfr_launch_jac_f32_512_recompute32(const MlpParams& P, int cap, int B, hipStream_t s);         // mlp_jac.hip (MODE 2 on 32-row tiles)
void sdfr_launch_fwd_f32_512_tile16(const MlpParams& P, int64_t n, hipStream_t s);                // mlp_jac.hip (forward on 16-row tiles: thin counted launches)
void sdfr_launch_jac_f16_512(const MlpParams& P, int cap, int B, hipStream_t s);                  // mlp_jac16.hip (mask-fed only)
void sdfr_launch_fwd_f16_512_tile16(const MlpParams& P, int64_t n, hipStream_t s);
void sdfr_launch_fwd_f16_512_tile64(const MlpParams& P, int64_t n, hipStream_t s);                // mlp_jac16.hip (half forward on 64-row tiles: the march's middle steps)                // mlp_jac16.hip (half forward on 16-row tiles)
void sdfr_launch_tail_f32_512(const MlpParams& P, int64_t n_rays, int spec_k, hipStream_t s);                      // mlp_jac.hip (MODE 4: sphere tracer's persistent tail)
void sdfr_launch_jac_f16_512_many(const MlpParams& P, int cap, int B, hipStream_t s);                                     // mlp_jac16.hip (32x32 tiles: many rows)
void sdfr_launch_tail_f16_512(const MlpParams& P, int64_t n_rays, int spec_k, hipStream_t s);                      // mlp_jac16.hip
void sdfr_launch_pool_f16_skip(const MlpParams& P, int64_t n, hipStream_t s);                                  // mlp_persist.hip (r06: pools of workgroups over the live tiles)
void sdfr_launch_pool_f16_ragged(const MlpParams& P, int64_t n, bool gather, hipStream_t s);
void sdfr_launch_pool_f16_ragged_half_tiles(const MlpParams& P, int64_t n, bool gather, hipStream_t s);
void sdfr_launch_pool_f16_ragged_quarter_tiles(const MlpParams& P, int64_t n, hipStream_t s);
void sdfr_launch_pool_f32_skip(const MlpParams& P, int64_t n, hipStream_t s);
void sdfr_launch_pool_f32_ragged(const MlpParams& P, int64_t n, bool gather, hipStream_t s);
void sdfr_launch_small(const MlpParams& P, int HP, int mode, int grid_x, int grid_y, hipStream_t s);   // mlp_small.hip (HP 128 / 256)
void sdfr_launch_ln(const MlpParams& P, int HP, bool jac, int grid_x, int grid_y, hipStream_t s);        // mlp_ln.hip (LayerNorm decoders)
int sdfr_ln_points_per_wg(int HP, bool jac);
