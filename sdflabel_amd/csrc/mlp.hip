// DeepSDF decoder: host side of the C ABI (weight packing, dispatch).  The kernels live in mlp_kernel.h and are instantiated in
// mlp_fwd32.hip / mlp_fwd16.hip / mlp_jac.hip / mlp_small.hip.
#include "mlp_kernel.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------
// Host side: packing + C ABI
// ---------------------------------------------------------------------------------------------------------------

// A decoder's weight images live on ONE device (sdfr_decoder_create's `device`); every launch below dereferences them from the stream it is
// given.  One process per GPU (sdflabel_amd/parallel.py) makes the current device, the stream's device and the decoder's device the same by
// construction -- a caller that mixes them (a handle created for cuda:0 used while cuda:1 is current) gets an error here instead of a fault or,
// with peer access enabled, silent cross-device weight reads over xGMI at a fraction of the HBM rate.
#define SDFR_DEVICE_CHECK(d, what)                                                                                                          \
    do {                                                                                                                                    \
        int cur_ = -1;                                                                                                                      \
        SDFR_HIP_CHECK(hipGetDevice(&cur_));                                                                                                \
        SDFR_REQUIRE(cur_ == (d)->device, "%s: the decoder's weights live on device %d but the current device (the launch stream's) is %d: " \
                     "create one decoder handle per device", what, (d)->device, cur_);                                                      \
    } while (0)

extern "C" int sdfr_decoder_create(sdfr_decoder** out, int n_lin, const int* in_dim, const int* out_dim,
                                   const int* inj_n, const int* inj_off, const float* const* h_W,
                                   const float* const* h_b, const float* const* h_ln_w, const float* const* h_ln_b, int n_inputs,
                                   int use_tanh, int device) {
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
    SdfrDeviceGuard dev_guard(device);              // allocations and copies below go to `device`; the caller's current device is restored
    SDFR_HIP_CHECK(dev_guard.err);

    sdfr_decoder* d = new sdfr_decoder();
    memset(d, 0, sizeof(*d));
    d->device = device; d->n_lin = n_lin; d->n_inputs = n_inputs; d->use_tanh = use_tanh; d->HP = HP;
    MlpParams& P = d->proto;
    P.n_mfma = n_lin - 1; P.n_inputs = n_inputs; P.use_tanh = use_tanh;
    int64_t off_f = 0, off_b = 0, off_h = 0, off_s = 0, off_bh = 0;
    const bool kinj = (n_inputs <= 8) && (HP == 512);
    P.kinj = kinj ? 1 : 0;
    for (int l = 0; l < n_lin; ++l) {
        MlpLayer& L = P.L[l];
        L.in_dim = in_dim[l]; L.out_dim = out_dim[l]; L.inj_n = inj_n[l]; L.inj_off = inj_off[l];
        L.ln = (l < n_lin - 1 && h_ln_w && h_ln_w[l] != nullptr) ? 1 : 0;
        if (L.ln) { SDFR_REQUIRE(h_ln_b && h_ln_b[l], "sdfr_decoder_create: LayerNorm weight without bias at layer %d", l); d->has_ln = 1; }
        L.kp_f = 16 * ((in_dim[l] + 15) / 16); L.kp_b = 16 * ((out_dim[l] + 15) / 16); L.kp_h = 32 * ((in_dim[l] + 31) / 32);
        L.off_f = (int)off_f; L.off_b = (int)off_b; L.off_h = (int)off_h;
        // half forward image: the re-injected input columns of a layer (latent_in / xyz_in_all) move behind the HP feature slots, k = HP +
        // input column, where the kernel keeps the tile's input rows (two more K tiles, zero padded): no operand patching between layers
        if (kinj && l > 0 && inj_n[l] > 0) L.kp_h = HP + 32;
        L.kp_s = 64 * ((in_dim[l] + 63) / 64); L.off_s = (int)off_s;
        L.kp_bh = 128 * ((out_dim[l] + 127) / 128); L.off_bh = (int)off_bh;
        d->macs += (int64_t)in_dim[l] * out_dim[l];
        if (l < n_lin - 1) { off_f += (int64_t)(L.kp_f / 4) * HP; off_b += (int64_t)(L.kp_b / 4) * HP; off_h += (int64_t)(L.kp_h / 8) * HP;
                             off_s += (int64_t)(L.kp_s / 8) * HP * 2; off_bh += (int64_t)(L.kp_bh / 8) * HP; }
    }
    // images: vector index [k / KV][row], KV consecutive k per 16-byte vector (KV = 4 floats or 8 halfs); zero padded
    std::vector<float> Wf((size_t)off_f * 4, 0.f), Wb((size_t)off_b * 4, 0.f), bias((size_t)(n_lin - 1) * HP, 0.f), wl(HP, 0.f);
    std::vector<_Float16> Wh((size_t)off_h * 8, (_Float16)0.f), Ws((size_t)off_s * 8, (_Float16)0.f), Wbh((size_t)off_bh * 8, (_Float16)0.f);
    for (int l = 0; l < n_lin - 1; ++l) {
        const MlpLayer& L = P.L[l];
        const float* W = h_W[l];
        for (int r = 0; r < L.out_dim; ++r)
            for (int k = 0; k < L.in_dim; ++k) {
                const float w = W[(size_t)r * L.in_dim + k];
                Wf[((size_t)L.off_f + (size_t)(k / 4) * HP + r) * 4 + (k % 4)] = w;
                const _Float16 wh = (_Float16)w;
                const int prev_out = l > 0 ? P.L[l - 1].out_dim : 0;
                const int kh = (kinj && l > 0 && L.inj_n > 0 && k >= prev_out) ? HP + L.inj_off + (k - prev_out) : k;
                Wh[((size_t)L.off_h + (size_t)(kh / 8) * HP + r) * 8 + (kh % 8)] = wh;
                const size_t es = ((size_t)L.off_s + ((size_t)(k / 8) * HP + r) * 2) * 8 + (k % 8);        // split forward: hi | lo
                Ws[es] = wh;
                Ws[es + 8] = (_Float16)((w - (float)wh) * 2048.f);
                Wb[((size_t)L.off_b + (size_t)(r / 4) * HP + k) * 4 + (r % 4)] = w;          // transposed: rows = in-features, k = out-features
                Wbh[((size_t)L.off_bh + (size_t)(r / 8) * HP + k) * 8 + (r % 8)] = wh;
            }
        for (int r = 0; r < L.out_dim; ++r) bias[(size_t)l * HP + r] = h_b[l][r];
    }
    for (int k = 0; k < in_dim[n_lin - 1]; ++k) wl[k] = h_W[n_lin - 1][k];
    P.b_last = h_b[n_lin - 1][0];

    SDFR_HIP_CHECK(hipMalloc(&d->d_Wf, Wf.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMalloc(&d->d_Wb, Wb.size() * sizeof(float)));
    if (d->has_ln) {
        std::vector<float> lg((size_t)(n_lin - 1) * HP, 0.f), lb((size_t)(n_lin - 1) * HP, 0.f);
        for (int l = 0; l < n_lin - 1; ++l)
            if (P.L[l].ln)
                for (int r = 0; r < P.L[l].out_dim; ++r) { lg[(size_t)l * HP + r] = h_ln_w[l][r]; lb[(size_t)l * HP + r] = h_ln_b[l][r]; }
        SDFR_HIP_CHECK(hipMalloc(&d->d_lng, lg.size() * sizeof(float)));
        SDFR_HIP_CHECK(hipMalloc(&d->d_lnb, lb.size() * sizeof(float)));
        SDFR_HIP_CHECK(hipMemcpy(d->d_lng, lg.data(), lg.size() * sizeof(float), hipMemcpyHostToDevice));
        SDFR_HIP_CHECK(hipMemcpy(d->d_lnb, lb.data(), lb.size() * sizeof(float), hipMemcpyHostToDevice));
        P.ln_gamma = d->d_lng; P.ln_beta = d->d_lnb;
    }
    SDFR_HIP_CHECK(hipMalloc(&d->d_Wh, Wh.size() * sizeof(_Float16)));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wh, Wh.data(), Wh.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMalloc(&d->d_Wbh, Wbh.size() * sizeof(_Float16)));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wbh, Wbh.data(), Wbh.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMalloc(&d->d_Ws, Ws.size() * sizeof(_Float16)));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Ws, Ws.data(), Ws.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMalloc(&d->d_bias, bias.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMalloc(&d->d_wlast, wl.size() * sizeof(float)));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wf, Wf.data(), Wf.size() * sizeof(float), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMemcpy(d->d_Wb, Wb.data(), Wb.size() * sizeof(float), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMemcpy(d->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    SDFR_HIP_CHECK(hipMemcpy(d->d_wlast, wl.data(), wl.size() * sizeof(float), hipMemcpyHostToDevice));
    P.Wf = d->d_Wf; P.Wb = d->d_Wb; P.Wh = d->d_Wh; P.Ws = d->d_Ws; P.Wbh = d->d_Wbh; P.bias = d->d_bias; P.w_last = d->d_wlast; P.fwd_np = 2;
    *out = d;
    return SDFR_OK;
}

// Debug: device buffer (2 * SDFR_MAX_LAYERS * 5 uint64) that forward kernels built with -DSDFR_MLP_TRACE fill with cycle stamps of their
// workgroup 0 (tools/cycle_trace.py); NULL (the default) disables it.  Not part of the renderer path.
static unsigned long long* g_trace = nullptr;
extern "C" int sdfr_debug_set_trace(void* device_buffer) { g_trace = (unsigned long long*)device_buffer; return SDFR_OK; }

extern "C" int sdfr_decoder_destroy(sdfr_decoder* d) {
    if (!d) return SDFR_OK;
    void* bufs[] = {d->d_Wf, d->d_Wb, d->d_Wh, d->d_Ws, d->d_Wbh, d->d_bias, d->d_wlast, d->d_lng, d->d_lnb, d->ln_ws};
    for (void* b : bufs) (void)hipFree(b);
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
    // HP bits per row and layer, rows rounded up to whole blocks of 128 (mask layout v2, mlp_kernel.h: sdfr_mask_dword; ft * nt * 2 = 128 * HP / 32)
    return ((n + 127) / 128) * 2 * (int64_t)(d->n_lin - 1) * ft * nt;
}

extern "C" int sdfr_mlp_forward(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf, "sdfr_mlp_forward: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward: n=%lld out of range", (long long)n);
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = mask_ws; P.trace = g_trace;
    const int grid = sdfr_cdiv(n, 64);
    if (d->has_ln) sdfr_launch_ln(P, d->HP, false, grid, 1, (hipStream_t)stream);        // LayerNorm decoders: no mask saving
    else if (d->HP == 512) sdfr_launch_fwd_f32_512(P, n, mask_ws != nullptr, (hipStream_t)stream);
    else sdfr_launch_small(P, d->HP, mask_ws ? 1 : 0, grid, 1, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

#ifndef SDFR_COUNTED_TILE16_ROWS
#define SDFR_COUNTED_TILE16_ROWS 4096      // 256 CUs x 16 rows: below this every 16-row tile has a CU of its own
#endif
// forward over the first *n_dev rows of `inputs` (n_dev: device int32, clamped to n_max = the launch bound): the per-step decoder call of the
// sphere tracer, whose active-ray count is produced on the device by the previous step (no host synchronisation).  half != 0: half operands.
extern "C" int sdfr_mlp_forward_counted(const sdfr_decoder* d, const float* inputs, int64_t n_max, const int32_t* n_dev, float* sdf, int half,
                                        void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && n_dev, "sdfr_mlp_forward_counted: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_counted");
    SDFR_REQUIRE(n_max >= 0 && n_max < (int64_t)1 << 31, "sdfr_mlp_forward_counted: n_max=%lld out of range", (long long)n_max);
    SDFR_REQUIRE(!half || (d->HP == 512 && !d->has_ln), "sdfr_mlp_forward_counted: half operands need a 512-wide decoder without LayerNorm");
    if (n_max == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n_max; P.sdf = sdf; P.maskbuf = nullptr; P.n_dev = n_dev; P.trace = g_trace;      // (cycle stamps: trace builds only)
    hipStream_t s = (hipStream_t)stream;
    if (half & 2) {
        // one product shape whatever the count (half | 2): 128-row tiles while they fill the chip, 64-row tiles of the SAME 32x32x16 products
        // below -- a row's value then has the same bits in every launch, so a crop marches identically alone and inside a batch (the 16-row
        // tiles below use 16x16x32 products, whose summation order differs)
        // (counts up to and INCLUDING 64 x 256 rows take the 64-row tiles: a launch bound of exactly that many rows -- the cone passes of a
        // 256x256 crop: 4096 cones x 4 samples -- then needs no 128-row launch at all)
        const int mid = 64 * 256 + 1;
        P.n_dev_lo = mid; P.n_dev_hi = 0x7fffffff;
        if (n_max >= mid) sdfr_launch_fwd_f16_512(P, n_max, false, s);
        P.n_dev_lo = 0; P.n_dev_hi = mid;
        sdfr_launch_fwd_f16_512_tile64(P, n_max < mid ? n_max : (int64_t)mid, s);
    } else if (half) {
        P.n_dev_lo = SDFR_COUNTED_TILE16_ROWS; P.n_dev_hi = 0x7fffffff;
        if (n_max >= SDFR_COUNTED_TILE16_ROWS) sdfr_launch_fwd_f16_512(P, n_max, false, s);
        P.n_dev_lo = 0; P.n_dev_hi = SDFR_COUNTED_TILE16_ROWS;
        sdfr_launch_fwd_f16_512_tile16(P, n_max < SDFR_COUNTED_TILE16_ROWS ? n_max : (int64_t)SDFR_COUNTED_TILE16_ROWS, s);
    } else if (d->has_ln) sdfr_launch_ln(P, d->HP, false, sdfr_cdiv(n_max, 64), 1, s);
    else if (d->HP == 512) {
        // two tile geometries, selected on the device by the count: 64-row tiles while the rows fill the chip, 16-row tiles (a quarter of the
        // latency per workgroup) below SDFR_COUNTED_TILE16_ROWS -- the launch that does not apply exits at once
        P.n_dev_lo = SDFR_COUNTED_TILE16_ROWS; P.n_dev_hi = 0x7fffffff;
        if (n_max >= SDFR_COUNTED_TILE16_ROWS) sdfr_launch_fwd_f32_512(P, n_max, false, s);
        P.n_dev_lo = 0; P.n_dev_hi = SDFR_COUNTED_TILE16_ROWS;
        sdfr_launch_fwd_f32_512_tile16(P, n_max < SDFR_COUNTED_TILE16_ROWS ? n_max : (int64_t)SDFR_COUNTED_TILE16_ROWS, s);
    } else sdfr_launch_small(P, d->HP, 0, sdfr_cdiv(n_max, 64), 1, s);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// sdfr_mlp_forward_f16 over the first *n_dev rows (device count, clamped to n_max), masks saved: the sphere tracer's hit pass in the decoder's
// own half precision (value + mask-fed half Jacobian at the hits: sdfr_mlp_jacobian with mask_from_f16 = 2)
extern "C" int sdfr_mlp_forward_f16_counted(const sdfr_decoder* d, const float* inputs, int64_t n_max, const int32_t* n_dev, float* sdf,
                                            uint32_t* mask_ws, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && n_dev, "sdfr_mlp_forward_f16_counted: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_f16_counted");
    SDFR_REQUIRE(n_max >= 0 && n_max < (int64_t)1 << 31, "sdfr_mlp_forward_f16_counted: n_max=%lld out of range", (long long)n_max);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_f16_counted: half operands need a 512-wide decoder without LayerNorm");
    if (n_max == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n_max; P.sdf = sdf; P.maskbuf = mask_ws; P.n_dev = n_dev; P.n_dev_lo = 0; P.n_dev_hi = 0; P.trace = nullptr;
    sdfr_launch_fwd_f16_512(P, n_max, mask_ws != nullptr, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// forward with float16 operands (f32 accumulate, f32 bias/ReLU/tanh): 128-point workgroup tiles
extern "C" int sdfr_mlp_forward_f16(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf, "sdfr_mlp_forward_f16: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_f16");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward_f16: n=%lld out of range", (long long)n);
    SDFR_REQUIRE(d->HP == 512, "sdfr_mlp_forward_f16: built for hidden widths 257..512 (padded width %d)", d->HP);
    SDFR_REQUIRE(!d->has_ln, "sdfr_mlp_forward_f16: LayerNorm decoders run in float32");
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = mask_ws; P.trace = g_trace;
    sdfr_launch_fwd_f16_512(P, n, mask_ws != nullptr, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// The exact-float32 forward with per-crop skip flags / over a ragged [B][rows_per_crop] row array: candidate reuse of the exact-f32 mode (r05;
// see the half versions below and csrc/surface.hip).  rows_per_crop of the ragged form a multiple of 64 (the f32 tile).
extern "C" int sdfr_mlp_forward_skip(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, const int32_t* skip, int64_t rows_per_crop,
                                     void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && skip && rows_per_crop > 0, "sdfr_mlp_forward_skip: bad argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_skip");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward_skip: n=%lld out of range", (long long)n);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_skip: 512-wide decoders without LayerNorm");
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = nullptr; P.skip = skip; P.skip_rows = rows_per_crop;
    P.n_crops = (int)((n + rows_per_crop - 1) / rows_per_crop);
    sdfr_launch_pool_f32_skip(P, n, (hipStream_t)stream);            // r06: a pool over the live tiles (same bits per row; nothing to do = one wave of dispatch)
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_mlp_forward_ragged(const sdfr_decoder* d, const float* inputs, int B, int64_t rows_per_crop, const int32_t* cnt, float* sdf,
                                       uint32_t* mask_ws, int half_tiles, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && cnt, "sdfr_mlp_forward_ragged: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_ragged");
    SDFR_REQUIRE(B >= 0 && rows_per_crop >= 0 && rows_per_crop % 64 == 0 && (int64_t)B * rows_per_crop < (int64_t)1 << 31,
                 "sdfr_mlp_forward_ragged: B=%d rows_per_crop=%lld (a multiple of 64, B * rows < 2^31)", B, (long long)rows_per_crop);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_ragged: 512-wide decoders without LayerNorm");
    if (B == 0 || rows_per_crop == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = (int64_t)B * rows_per_crop; P.sdf = sdf; P.maskbuf = mask_ws; P.crop_cnt = cnt; P.crop_rows = rows_per_crop; P.trace = nullptr;
    // (half-size tiles exist for the float16 kernel only: a 32-row float32 instantiation measured 26 % faster at one crop, but the float32
    // kernel's last linear is summed in NT / PT slices per point -- 8 on 64-row tiles, 16 on 32-row tiles -- so its values differ in the last
    // bits from the 64-row launch's; the half kernel fixes that partition at 4 for every tile size (mlp_kernel.h, "last linear").  Not shipped)
    if (half_tiles) { sdfr_set_error("sdfr_mlp_forward_ragged: half_tiles is a float16 option (sdfr_mlp_forward_f16_ragged)"); return SDFR_E_UNSUPPORTED; }
    P.n_crops = B;
    sdfr_launch_pool_f32_ragged(P, P.n, false, (hipStream_t)stream);  // r06: a pool of workgroups over the live tiles (same bits per row)
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// sdfr_mlp_forward_f16 over a ragged [B][rows_per_crop] row array (rows_per_crop a multiple of 128): crop b's first cnt[b] rows are evaluated
// (whole 128-row tiles: the rows up to the next multiple of 128 are computed too and must be readable), masks saved in the forward layout of
// a B * rows_per_crop-row launch.  The candidate pass of the float16 reuse mode (BatchRenderer, decoder.candidate_reuse): a row's value and
// masks have the bits sdfr_mlp_forward_f16 gives that row in any launch.
extern "C" int sdfr_mlp_forward_f16_ragged(const sdfr_decoder* d, const float* inputs, int B, int64_t rows_per_crop, const int32_t* cnt, float* sdf,
                                           uint32_t* mask_ws, int half_tiles, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && cnt, "sdfr_mlp_forward_f16_ragged: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_f16_ragged");
    SDFR_REQUIRE(B >= 0 && rows_per_crop >= 0 && rows_per_crop % 128 == 0 && (int64_t)B * rows_per_crop < (int64_t)1 << 31,
                 "sdfr_mlp_forward_f16_ragged: B=%d rows_per_crop=%lld (a multiple of 128, B * rows < 2^31)", B, (long long)rows_per_crop);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_f16_ragged: half operands need a 512-wide decoder without LayerNorm");
    if (B == 0 || rows_per_crop == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = (int64_t)B * rows_per_crop; P.sdf = sdf; P.maskbuf = mask_ws; P.crop_cnt = cnt; P.crop_rows = rows_per_crop; P.trace = nullptr;
    P.n_crops = B;
    if (half_tiles) sdfr_launch_pool_f16_ragged_half_tiles(P, P.n, false, (hipStream_t)stream);       // 64-row tiles
    else sdfr_launch_pool_f16_ragged(P, P.n, false, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// The candidates' pass of the reuse modes WITHOUT the gathered copy (r06): row s < cnt[b] of crop b is inputs[b * rows_per_crop_in + cidx[b][s]]
// (cidx int32 [B][stride], stride a multiple of the tile: 128, or 64 with half_tiles / float32); values -> sdf [B][stride], masks -> mask_ws in
// the forward layout of a B * stride-row launch -- exactly what sdfr_candidate_rows + sdfr_mlp_forward(_f16)_ragged produce, one launch less.
// half != 0: the float16 kernel (half_tiles = 1: 64-row tiles, 2: 32-row tiles); half == 0: the exact-float32 kernel.
extern "C" int sdfr_mlp_forward_candidates(const sdfr_decoder* d, const float* inputs, int64_t rows_per_crop_in, int B, const int32_t* cidx,
                                           int64_t stride, const int32_t* cnt, float* sdf, uint32_t* mask_ws, int half, int half_tiles,
                                           void* stream) {
    SDFR_REQUIRE(d && inputs && cidx && cnt && sdf, "sdfr_mlp_forward_candidates: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_candidates");
    const int tile = half ? (half_tiles == 2 ? 32 : (half_tiles ? 64 : 128)) : 64;
    SDFR_REQUIRE(B >= 0 && stride >= 0 && stride % tile == 0 && (int64_t)B * stride < (int64_t)1 << 31 && rows_per_crop_in > 0 &&
                 (int64_t)B * rows_per_crop_in < (int64_t)1 << 31, "sdfr_mlp_forward_candidates: B=%d stride=%lld (a multiple of %d) rows_per_crop_in=%lld",
                 B, (long long)stride, tile, (long long)rows_per_crop_in);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_candidates: 512-wide decoders without LayerNorm");
    SDFR_REQUIRE(half || !half_tiles, "sdfr_mlp_forward_candidates: half_tiles is a float16 option");
    if (B == 0 || stride == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = (int64_t)B * stride; P.sdf = sdf; P.maskbuf = mask_ws; P.crop_cnt = cnt; P.crop_rows = stride; P.trace = nullptr;
    P.gather_idx = cidx; P.gather_rows = rows_per_crop_in; P.n_crops = B;
    if (!half) sdfr_launch_pool_f32_ragged(P, P.n, true, (hipStream_t)stream);
    else if (half_tiles == 2) sdfr_launch_pool_f16_ragged_quarter_tiles(P, P.n, (hipStream_t)stream);
    else if (half_tiles) sdfr_launch_pool_f16_ragged_half_tiles(P, P.n, true, (hipStream_t)stream);
    else sdfr_launch_pool_f16_ragged(P, P.n, true, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// the half pass of the two-stage evaluation with per-crop skip flags (device int32[B], rows_per_crop rows each): flagged crops are not
// evaluated (their rows of `sdf` keep the previous values) -- see sdfr_prefilter_plan
extern "C" int sdfr_mlp_forward_f16_skip(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, const int32_t* skip,
                                         int64_t rows_per_crop, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && skip && rows_per_crop > 0, "sdfr_mlp_forward_f16_skip: bad argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_f16_skip");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward_f16_skip: n=%lld out of range", (long long)n);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_f16_skip: 512-wide decoders without LayerNorm");
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = nullptr; P.skip = skip; P.skip_rows = rows_per_crop;
    P.n_crops = (int)((n + rows_per_crop - 1) / rows_per_crop);
    sdfr_launch_pool_f16_skip(P, n, (hipStream_t)stream);            // r06: a pool over the live tiles
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// forward with error-compensated float16 operands: every float32 operand x is carried as the pair  hi = half(x),
// lo = half((x - hi) * 2^11)  (22 significand bits) and each product as  hi*hi + (hi*lo + lo*hi) * 2^-11  on the f16 matrix cores with
// float32 accumulation -- three f16 MFMAs replace sixteen f32 MFMA passes.  Results agree with sdfr_mlp_forward to float32 rounding
// noise (the same order as a change of summation order); masks are saved in the f32 forward's layout.
extern "C" int sdfr_mlp_forward_split(const sdfr_decoder* d, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream) {
    SDFR_REQUIRE(d && inputs && sdf, "sdfr_mlp_forward_split: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_split");
    SDFR_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "sdfr_mlp_forward_split: n=%lld out of range", (long long)n);
    SDFR_REQUIRE(d->HP == 512, "sdfr_mlp_forward_split: built for hidden widths 257..512 (padded width %d)", d->HP);
    SDFR_REQUIRE(!d->has_ln, "sdfr_mlp_forward_split: LayerNorm decoders run in float32");
    if (n == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n; P.sdf = sdf; P.maskbuf = mask_ws; P.trace = g_trace;
    sdfr_launch_fwd_split_512(P, n, mask_ws != nullptr, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// ... over the first *n_dev rows only (device count, clamped to n_max): float32-grade values of a row list whose length lives on the device
// (the audit of the two-stage evaluation, csrc/surface.hip)
extern "C" int sdfr_mlp_forward_split_counted(const sdfr_decoder* d, const float* inputs, int64_t n_max, const int32_t* n_dev, float* sdf,
                                              void* stream) {
    SDFR_REQUIRE(d && inputs && sdf && n_dev, "sdfr_mlp_forward_split_counted: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_forward_split_counted");
    SDFR_REQUIRE(n_max >= 0 && n_max < (int64_t)1 << 31, "sdfr_mlp_forward_split_counted: n_max=%lld out of range", (long long)n_max);
    SDFR_REQUIRE(d->HP == 512 && !d->has_ln, "sdfr_mlp_forward_split_counted: built for 512-wide decoders without LayerNorm");
    if (n_max == 0) return SDFR_OK;
    MlpParams P = d->proto;
    P.inputs = inputs; P.n = n_max; P.sdf = sdf; P.maskbuf = nullptr; P.n_dev = n_dev; P.n_dev_lo = 0; P.n_dev_hi = 0; P.trace = nullptr;
    sdfr_launch_fwd_split_512(P, n_max, false, (hipStream_t)stream);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

extern "C" int sdfr_mlp_jacobian(const sdfr_decoder* d, const float* inputs, int64_t rows_per_crop, int B,
                                 const int32_t* idx, int cap, const int32_t* cnt, float* J, float* sdf_sel,
                                 const float* sdf_full, const uint32_t* mask_ws, int mask_from_f16, void* stream) {
    SDFR_REQUIRE(d && inputs && idx && J, "sdfr_mlp_jacobian: NULL argument");
    SDFR_DEVICE_CHECK(d, "sdfr_mlp_jacobian");
    SDFR_REQUIRE(B >= 0 && cap >= 0, "sdfr_mlp_jacobian: negative size");
    SDFR_REQUIRE(mask_ws == nullptr || sdf_full != nullptr, "sdfr_mlp_jacobian: mask_ws needs sdf_full");
    if (B == 0 || cap == 0) return SDFR_OK;
    hipStream_t s = (hipStream_t)stream;
    MlpParams P = d->proto;
    P.inputs = inputs; P.rows_per_crop = rows_per_crop; P.idx = idx; P.cnt = cnt; P.cap = cap; P.J = J; P.sdf_sel = sdf_sel;
    P.sdf_in = sdf_full; P.maskbuf = const_cast<uint32_t*>(mask_ws); P.trace = g_trace;      // (cycle stamps: trace builds only)
    P.n_crops = B;                                                   // (the pool geometry of the half kernel walks the crops' live band tiles)
    const bool many_rows = (mask_from_f16 & SDFR_JAC_MANY_ROWS) != 0;       // hint: far more rows than 16 x the CU count (recomputing kernel on 32-row tiles)
    const bool half_tiles = (mask_from_f16 & SDFR_JAC_HALF_TILES) != 0;     // masks saved by a half-size-tile forward (sdfr_mlp_forward*_ragged, half_tiles = 1)
    const bool quarter_tiles = (mask_from_f16 & SDFR_JAC_QUARTER_TILES) != 0;
    mask_from_f16 &= ~(SDFR_JAC_MANY_ROWS | SDFR_JAC_HALF_TILES | SDFR_JAC_QUARTER_TILES);
    P.fwd_np = mask_from_f16 ? sdfr_fwd_f16_512_np() : (d->HP == 512 ? sdfr_fwd_f32_512_np() : 2);
    if (half_tiles) P.fwd_np /= 2;
    if (quarter_tiles) P.fwd_np /= 4;
    SDFR_REQUIRE(P.fwd_np >= 1, "sdfr_mlp_jacobian: tile-size flags do not fit the forward geometry");
    SDFR_REQUIRE(mask_from_f16 >= 0 && mask_from_f16 <= 2, "sdfr_mlp_jacobian: mask_from_f16 = %d (0, 1 or 2)", mask_from_f16);
    // masks saved by the forward launch make the recomputation unnecessary (not for use_tanh decoders: their output
    // derivative needs the pre-tanh value)
    const bool from_masks = mask_ws && sdf_full && !d->use_tanh && !d->has_ln;
    if (d->has_ln) {
        // recomputing Jacobian; the normalised pre-activations of every layer are spilled to a scratch owned by the decoder handle
        const int pt = sdfr_ln_points_per_wg(d->HP, true);
        const int gx = sdfr_cdiv(cap, pt);
        const size_t need = (size_t)gx * B * (d->n_lin - 1) * d->HP * pt * sizeof(float);
        if (need > d->ln_ws_bytes) {
            SDFR_HIP_CHECK(hipStreamSynchronize(s));
            if (d->ln_ws) SDFR_HIP_CHECK(hipFree(d->ln_ws));
            d->ln_ws = nullptr; d->ln_ws_bytes = 0;
            SDFR_HIP_CHECK(hipMalloc(&d->ln_ws, need));
            d->ln_ws_bytes = need;
        }
        P.ln_ws = d->ln_ws;
        sdfr_launch_ln(P, d->HP, true, gx, B, s);
    } else if (d->HP == 512 && mask_from_f16 == 2 && from_masks && many_rows) sdfr_launch_jac_f16_512_many(P, cap, B, s);
    else if (d->HP == 512 && mask_from_f16 == 2 && from_masks) sdfr_launch_jac_f16_512(P, cap, B, s);      // half operands, like the forward
    else if (d->HP == 512 && !from_masks && many_rows) sdfr_launch_jac_f32_512_recompute32(P, cap, B, s);
    else if (d->HP == 512) sdfr_launch_jac_f32_512(P, cap, B, from_masks, s);
    else sdfr_launch_small(P, d->HP, from_masks ? 3 : 2, sdfr_cdiv(cap, 32), B, s);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
