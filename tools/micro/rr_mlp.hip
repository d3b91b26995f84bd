// Micro-benchmark / prototype (NOT part of libsdfr_hip.so): a chain of NL full-width half layers  y = relu(W x + b), W [512][512] f16,
// f32 accumulate, evaluated by a REGISTER-RESIDENT kernel -- the design DESIGN.md section 8 (2a) argues for:
//   * 4 waves per workgroup (one per SIMD, 512 registers each), every wave owns 32 points and ALL 512 features: 16 accumulator tiles of
//     v_mfma_f32_32x32x16_f16 (256 accumulator registers);
//   * a point's activations never leave its wave: after bias + ReLU + f32->f16 the accumulator tile of features 32f .. 32f+31 becomes the
//     B fragments of the next layer's k tiles 2f and 2f+1 by one exchange between lanes p and p+32 (lane (p,g) of a tile holds features
//     8i + 4g + j; the B fragment of lane (p,g) wants features 8g .. 8g+7 of a 16-feature k tile);
//   * the weights are the only shared operand: every stage (16 k x 512 features = 16 KiB) is staged ONCE per CU into an LDS ring by
//     global_load_lds_dwordx4 (each wave a quarter), published by a counted s_waitcnt vmcnt + one s_barrier per stage, and read by all four
//     waves as A fragments; the ring streams across layer boundaries.
// No LDS round trip of activations, no barrier between layers.  Checked here against a naive kernel on the first 256 points.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rr_mlp tools/micro/rr_mlp.hip && /tmp/rr_mlp [points = 512000] [layers = 8]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HP 512            // layer width (features = k)
#define KT 16             // k per stage (one 32x32x16 MFMA k tile)
#define NKT (HP / KT)     // 32 stages per layer
#define FTILES (HP / 32)  // 16 feature tiles per wave
#ifndef RING
#define RING 8            // LDS ring slots of 16 KiB (NKT % RING == 0); RING - 2 stages are in flight while one is consumed
#endif
#ifndef BARRIER_BUILTIN
#define BARRIER_BUILTIN 0
#endif
#ifndef DMA_TOP
#define DMA_TOP 0
#endif
#ifndef GLDS_IMM
#define GLDS_IMM 0
#endif
#ifndef INPLACE_A
#define INPLACE_A 0
#endif
#ifndef PERMLANE_SWAP
#define PERMLANE_SWAP 1
#endif
#ifndef KSUB
#define KSUB 1            // k tiles (of 16) per ring stage: one barrier, one DMA batch and one drain of the LDS read queue per KSUB * 512 matrix cycles
#endif
#define TILE_VEC (2 * HP)             // 16-byte vectors per k tile: [lane group g][feature]
#define STAGE_VEC (KSUB * TILE_VEC)   // ... per ring stage

// weight image: Wimg[l][t][g][f] = 8 halves W[l][f][16 t + 8 g .. + 7]   (A fragment of lane (f % 32, g) of feature tile f / 32)
// bias image: [l][HP] float

#ifndef GLDS_ASM
#define GLDS_ASM 0
#endif
__device__ __forceinline__ void glds16(const void* g, void* lds) {
#if GLDS_ASM
    // (the builtin makes hipcc drain the LDS read queue -- s_waitcnt lgkmcnt(0) -- at every M0 write that follows a DMA; in asm the wave-uniform
    // LDS base goes to M0 by hand and the compiler sees neither the DMA nor the M0 write)
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m) : "memory", "m0");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
#endif
}

#ifndef ASM_KLOOP
#define ASM_KLOOP 0
#endif
#if ASM_KLOOP
// One half k tile in inline asm: 8 products on fragments u0..u7 (read by the PREVIOUS block) while 8 fragments n0..n7 are read for the NEXT block.
// The LDS reads are invisible to the compiler, so none of its s_waitcnt lgkmcnt(0) drains appear; the waits are counted by hand: before product f the
// reads still allowed in flight are u(f+1)..u7 and n0..nf = 8, whatever f (older compiler-issued LDS operations only make the wait stricter: the
// queue is in order).  OFF = byte offset of the block's first fragment from `ad` (fragments 512 B apart).  s_nop: the hazard recogniser does not see
// the MFMAs either (accumulator written by v_accvgpr_write before / read by v_accvgpr_read after the block).
#define RR_STEP(F, OFFB)                                                        \
    "ds_read_b128 %[n" #F "], %[ad] offset:" #OFFB "\n\t"                      \
    "s_waitcnt lgkmcnt(8)\n\t"                                                  \
    "v_mfma_f32_32x32x16_f16 %[c" #F "], %[u" #F "], %[b], %[c" #F "]\n\t"
#define RR_HALF(ACC, U, N, BFRAG, AD, O0, O1, O2, O3, O4, O5, O6, O7)                                                                  \
    asm volatile("s_nop 7\n\t" RR_STEP(0, O0) RR_STEP(1, O1) RR_STEP(2, O2) RR_STEP(3, O3) RR_STEP(4, O4) RR_STEP(5, O5) RR_STEP(6, O6) RR_STEP(7, O7) \
                 "s_nop 7\n\ts_nop 7"                                                                                                    \
                 : [c0] "+a"(ACC[0]), [c1] "+a"(ACC[1]), [c2] "+a"(ACC[2]), [c3] "+a"(ACC[3]), [c4] "+a"(ACC[4]), [c5] "+a"(ACC[5]),      \
                   [c6] "+a"(ACC[6]), [c7] "+a"(ACC[7]), [n0] "=&v"(N[0]), [n1] "=&v"(N[1]), [n2] "=&v"(N[2]), [n3] "=&v"(N[3]),          \
                   [n4] "=&v"(N[4]), [n5] "=&v"(N[5]), [n6] "=&v"(N[6]), [n7] "=&v"(N[7])                                                 \
                 : [u0] "v"(U[0]), [u1] "v"(U[1]), [u2] "v"(U[2]), [u3] "v"(U[3]), [u4] "v"(U[4]), [u5] "v"(U[5]), [u6] "v"(U[6]),         \
                   [u7] "v"(U[7]), [b] "v"(BFRAG), [ad] "v"(AD)                                                                           \
                 : "memory")
#endif

template <int NL_MAX>
__global__ __launch_bounds__(256, 1) void rr_mlp_kernel(const h16x8* __restrict__ Wimg, const float* __restrict__ bias, const h16x8* __restrict__ x,
                                                        h16x8* __restrict__ out, int n_points, int NL) {
    __shared__ h16x8 ring[RING * STAGE_VEC];              // 64 KiB
    __shared__ float lbias[NL_MAX * HP];                  // 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, g = lane >> 5;
    const int64_t pt = (int64_t)blockIdx.x * 128 + wave * 32 + p;
    const int total = NL * NKT / KSUB;                    // ring stages of the whole chain

    // this wave's quarter of a stage: vectors [wave * 256 + i * 64 + lane], i = 0 .. 3.  Stages beyond the last are clamped to it: the load then
    // refills a slot nobody reads any more, and the loop body needs no branch (one basic block per layer: the compiler's own waits stay exact)
    auto issue = [&](int S) {
#ifdef ABL_NO_DMA
        if (S >= RING - 1) return;                          // ablation (timing only, wrong results): the ring is filled once and never refilled
#endif
        const int Sc = S < total ? S : total - 1;
        const h16x8* src = Wimg + (int64_t)Sc * STAGE_VEC + wave * (256 * KSUB) + lane;
        h16x8* dst = ring + (S % RING) * STAGE_VEC + wave * (256 * KSUB);     // wave-uniform base; the hardware adds lane * 16
#if GLDS_IMM
        // one M0 value per stage: the instruction's immediate offset moves the global AND the LDS address of the other three quarters-of-a-quarter
        const __attribute__((address_space(1))) void* gs = (const __attribute__((address_space(1))) void*)src;
        __attribute__((address_space(3))) void* ls = (__attribute__((address_space(3))) void*)dst;
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 3072, 0);
#else
#pragma unroll
        for (int i = 0; i < 4 * KSUB; ++i) glds16(src + i * 64, dst + i * 64);
#endif
    };
#pragma unroll
    for (int S = 0; S < RING - 1; ++S) issue(S);
    for (int e = tid; e < NL * HP; e += 256) lbias[e] = bias[e];

    // layer-0 operand: B fragment of k tile t = x[pt][16 t + 8 g .. + 7]
    h16x8 B[NKT];
    const bool live = pt < n_points;
#pragma unroll
    for (int t = 0; t < NKT; ++t) B[t] = live ? x[pt * (HP / 8) + 2 * t + g] : (h16x8)(h16)0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the x loads; also lands the first RING - 1 stages -- once)
    __syncthreads();

    const h16x8* lds0 = ring + g * HP + p;
#if INPLACE_A
    // (variant, measured slower: 1047 against 1112 TFLOP/s) ONE buffer of a whole stage, refilled in place, the barrier at the stage boundary
    h16x8 A[FTILES];
#pragma unroll
    for (int f = 0; f < FTILES; ++f) A[f] = lds0[f * 32];
#else
    // A fragments: two half-stage buffers of 8 feature tiles; half 0 of stage 0 is read here, every later half under the other half's products
    h16x8 A0[8], A1[8];
#if ASM_KLOOP
    const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(ring + g * HP + p);     // LDS byte address of this lane's fragment column
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:512\n\tds_read_b128 %2, %8 offset:1024\n\tds_read_b128 %3, %8 offset:1536\n\t"
                 "ds_read_b128 %4, %8 offset:2048\n\tds_read_b128 %5, %8 offset:2560\n\tds_read_b128 %6, %8 offset:3072\n\tds_read_b128 %7, %8 offset:3584"
                 : "=&v"(A0[0]), "=&v"(A0[1]), "=&v"(A0[2]), "=&v"(A0[3]), "=&v"(A0[4]), "=&v"(A0[5]), "=&v"(A0[6]), "=&v"(A0[7])
                 : "v"(lds_b)
                 : "memory");
#else
#pragma unroll
    for (int f = 0; f < 8; ++f) A0[f] = lds0[f * 32];
#endif
#endif

    for (int l = 0; l < NL; ++l) {
        f32x16 acc[FTILES];
#pragma unroll
        for (int f = 0; f < FTILES; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
        static_assert((NKT / KSUB) % RING == 0 && NKT % KSUB == 0, "the ring slot of a k tile must depend on the tile's index in the layer alone");
        static_assert(KSUB == 1 || (!INPLACE_A && !DMA_TOP), "the variants exist for one k tile per stage only");
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            const int S = l * (NKT / KSUB) + t / KSUB;                          // ring stage of k tile t
            const h16x8* slot = lds0 + ((t / KSUB) % RING) * STAGE_VEC + (t % KSUB) * TILE_VEC;
            const int tn = (t + 1) % NKT;                                       // (the first k tile of the next layer sits in ring slot 0 again)
            const h16x8* next = lds0 + ((tn / KSUB) % RING) * STAGE_VEC + (tn % KSUB) * TILE_VEC;
#if INPLACE_A
            (void)slot;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING - 3) * 4) : "memory");
            issue(S + RING - 1);
#pragma unroll
            for (int f = 0; f < FTILES; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[f], B[t], acc[f], 0, 0, 0);
                A[f] = next[f * 32];
            }
#pragma unroll
            for (int f = 0; f < FTILES; ++f) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#else
#if DMA_TOP
            // refill at the TOP of the stage, where the compiler's drain of the LDS read queue (it comes with every M0 write after a DMA) only waits
            // for fragments this half needs anyway: the slot of stage S - 2 is free since the barrier in the middle of stage S - 1
            issue(S + RING - 2);                            // (S = 0 re-issues the prologue's last stage into its own slot: harmless, and no branch)
#endif
            // feature tiles 0 .. 7 of stage S (fragments read during the previous half) while tiles 8 .. 15 are read
#if ASM_KLOOP
            const uint32_t ad_cur = lds_b + (uint32_t)((((t / KSUB) % RING) * STAGE_VEC + (t % KSUB) * TILE_VEC) * 16);
            const uint32_t ad_nxt = lds_b + (uint32_t)((((tn / KSUB) % RING) * STAGE_VEC + (tn % KSUB) * TILE_VEC) * 16);
            RR_HALF((acc + 0), A0, A1, B[t], ad_cur, 4096, 4608, 5120, 5632, 6144, 6656, 7168, 7680);
#else
#pragma unroll
            for (int f = 0; f < 8; ++f) A1[f] = slot[(8 + f) * 32];
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[f], B[t], acc[f], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 8; ++f) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#endif
            // stage S + 1 has landed (this wave's quarter: all but the RING - 3 younger stages' loads are done), then everybody's; every wave has
            // also finished the products of stage S - 1, so that stage's slot is free: refill it with stage S + RING - 1
            if (t % KSUB == KSUB - 1) {                     // (compile-time: t is an unrolled index) the stage's last k tile: publish the next stage
#if BARRIER_BUILTIN
                {   // (variant: the compiler's own builtins instead of inline asm -- it can then count the LDS reads across the barrier)
                    constexpr int N = (RING - 3) * 4 * KSUB;
                    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
                    __builtin_amdgcn_s_barrier();
#if BARRIER_BUILTIN == 1
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       // (2: without the fence -- the check decides whether that is safe)
#endif
                }
#else
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING - 3) * 4 * KSUB) : "memory");
#endif
#if !DMA_TOP
                issue(S + RING - 1);
#endif
            }
            // feature tiles 8 .. 15 of k tile t while tiles 0 .. 7 of k tile t + 1 are read
#if ASM_KLOOP
            RR_HALF((acc + 8), A1, A0, B[t], ad_nxt, 0, 512, 1024, 1536, 2048, 2560, 3072, 3584);
#else
#pragma unroll
            for (int f = 0; f < 8; ++f) A0[f] = next[f * 32];
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[8 + f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[f], B[t], acc[8 + f], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 8; ++f) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); }
#endif
#endif
        }
        // epilogue in registers: bias + ReLU + pack; accumulator tile f -> B fragments of k tiles 2f, 2f + 1
        const float* bl = lbias + l * HP;
#pragma unroll
        for (int f = 0; f < FTILES; ++f) {
            uint32_t P[4][2];                               // [i][pair]: features 32 f + 8 i + 4 g + {0,1}, {2,3} as packed halves
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 b4 = *reinterpret_cast<const float4*>(bl + 32 * f + 8 * i + 4 * g);
                f32x2 lo = {acc[f][4 * i + 0] + b4.x, acc[f][4 * i + 1] + b4.y}, hi = {acc[f][4 * i + 2] + b4.z, acc[f][4 * i + 3] + b4.w};
                h16x2 l2 = __builtin_convertvector(lo, h16x2), h2 = __builtin_convertvector(hi, h16x2);
                const h16x2 z = {(h16)0, (h16)0};
                l2 = __builtin_elementwise_max(l2, z);
                h2 = __builtin_elementwise_max(h2, z);
                P[i][0] = *reinterpret_cast<uint32_t*>(&l2);
                P[i][1] = *reinterpret_cast<uint32_t*>(&h2);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {                   // k tile 2 f + h: features 16 h + 8 g .. + 7 = (i = 2h, both groups) for g = 0, (i = 2h + 1, both groups) for g = 1
                uint32_t w[4];
#if PERMLANE_SWAP
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // v_permlane32_swap X, Y: lanes 32..63 of X <-> lanes 0..31 of Y.  With X = (i = 2h) and Y = (i = 2h + 1): afterwards X holds
                    // features 8g + 0..3 and Y features 8g + 4..7 of the k tile in BOTH lane groups -- the exchange without LDS and without a wait
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 r = __builtin_amdgcn_permlane32_swap(P[2 * h][q], P[2 * h + 1][q], false, false);
                    w[q] = r[0];
                    w[2 + q] = r[1];
                }
#else
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // lane (p,0) keeps (i = 2h, g = 0) and needs (i = 2h, g = 1); lane (p,1) keeps (i = 2h + 1, g = 1) and needs (i = 2h + 1, g = 0):
                    // each sends what the other needs (v_permlane32_swap does the same without the crossbar; ds_bpermute here for clarity)
                    const uint32_t send = g ? P[2 * h][q] : P[2 * h + 1][q];
                    const uint32_t recv = (uint32_t)__shfl_xor((int)send, 32, 64);
                    w[q] = g ? recv : P[2 * h][q];          // features 8 g + 0 .. 3
                    w[2 + q] = g ? P[2 * h + 1][q] : recv;  // features 8 g + 4 .. 7
                }
#endif
                uint32_t* dst = reinterpret_cast<uint32_t*>(&B[2 * f + h]);
                dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clamped refills of the last stages: nothing may land in LDS after the workgroup has gone)
    if (live) {
#pragma unroll
        for (int t = 0; t < NKT; ++t) out[pt * (HP / 8) + 2 * t + g] = B[t];
    }
}

// naive reference: one thread per (point, feature) and layer
__global__ void ref_layer(const h16* __restrict__ W, const float* __restrict__ b, const h16* __restrict__ xin, h16* __restrict__ xout, int n) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x, pt = blockIdx.y;
    if (f >= HP || pt >= n) return;
    float s = 0.f;
    for (int k = 0; k < HP; ++k) s += (float)W[(size_t)f * HP + k] * (float)xin[(size_t)pt * HP + k];
    s += b[f];
    xout[(size_t)pt * HP + f] = (h16)(s > 0.f ? s : 0.f);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512000;
    const int NL = argc > 2 ? atoi(argv[2]) : 8;
    if (NL < 1 || NL > 8 || n < 256) { printf("layers 1 .. 8, points >= 256\n"); return 1; }
    std::vector<h16> W((size_t)NL * HP * HP), X((size_t)n * HP);
    std::vector<float> Bv((size_t)NL * HP);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    const float amp = argc > 3 ? (float)atof(argv[3]) : 1.f;    // 0: all-zero operands (what the matrix pipes' data-dependent power draw costs: timing only)
    for (auto& w : W) w = (h16)(rnd() * 0.125f * amp);      // |W x| stays O(1) over 8 layers
    for (auto& b : Bv) b = rnd() * 0.2f * amp;
    for (auto& x : X) x = (h16)(rnd() * 2.0f * amp);
    std::vector<h16> Wimg(W.size());
    for (int l = 0; l < NL; ++l)
        for (int t = 0; t < NKT; ++t)
            for (int g = 0; g < 2; ++g)
                for (int f = 0; f < HP; ++f)
                    for (int j = 0; j < 8; ++j)
                        Wimg[((((size_t)l * NKT + t) * 2 + g) * HP + f) * 8 + j] = W[((size_t)l * HP + f) * HP + 16 * t + 8 * g + j];
    h16 *dW, *dWimg, *dX, *dOut, *dR0, *dR1;
    float* dB;
    CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dWimg, W.size() * 2)); CK(hipMalloc(&dX, X.size() * 2)); CK(hipMalloc(&dOut, X.size() * 2));
    CK(hipMalloc(&dR0, 256 * HP * 2)); CK(hipMalloc(&dR1, 256 * HP * 2)); CK(hipMalloc(&dB, Bv.size() * 4));
    CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dWimg, Wimg.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bv.data(), Bv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dOut, 0, X.size() * 2));
    const int grid = (n + 127) / 128;
    auto launch = [&]() {
        hipLaunchKernelGGL(rr_mlp_kernel<8>, dim3(grid), dim3(256), 0, 0, (const h16x8*)dWimg, dB, (const h16x8*)dX, (h16x8*)dOut, n, NL);
    };
    launch();
    CK(hipDeviceSynchronize());
    // reference on the first 256 points
    CK(hipMemcpy(dR0, dX, 256 * HP * 2, hipMemcpyDeviceToDevice));
    for (int l = 0; l < NL; ++l) {
        hipLaunchKernelGGL(ref_layer, dim3(HP / 256, 256), dim3(256), 0, 0, dW + (size_t)l * HP * HP, dB + l * HP, dR0, dR1, 256);
        std::swap(dR0, dR1);
    }
    CK(hipDeviceSynchronize());
    std::vector<h16> got(256 * HP), want(256 * HP), last((size_t)128 * HP);
    CK(hipMemcpy(got.data(), dOut, got.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(want.data(), dR0, want.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(last.data(), dOut + (size_t)(n - 128) * HP, last.size() * 2, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0, sum = 0; int nz = 0;
    for (size_t i = 0; i < got.size(); ++i) {
        const double a = (double)got[i], b = (double)want[i];
        maxd = fmax(maxd, fabs(a - b)); maxv = fmax(maxv, fabs(b)); sum += b; nz += b != 0.0;
    }
    double lsum = 0; for (auto v : last) lsum += (double)v;
    printf("check on 256 points x %d features after %d layers: max |rr - naive| = %.4g (max |value| %.4g, %d non-zero, sum %.6g); last 128 points sum %.6g\n",
           HP, NL, maxd, maxv, nz, sum, lsum);
    const bool ok = maxd <= 2e-2 * fmax(1.0, maxv) && (nz > 1000 || amp == 0.f);
    printf(ok ? "CHECK OK\n" : "CHECK FAILED\n");
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms / 5);
    }
    const double flop = 2.0 * (double)n * NL * HP * HP;
    printf("%d points, %d layers of 512 x 512: %.4f ms per launch = %.0f TFLOP/s = %.1f %% of 2.5 PFLOP/s  (%.1f us per 64 000 points)\n", n, NL, best,
           flop / (best * 1e-3) / 1e12, 100.0 * flop / (best * 1e-3) / 2.5e15, best * 1e3 * 64000.0 / n);
    return ok ? 0 : 2;
}
