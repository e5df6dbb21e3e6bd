// Fused conv + BN + ReLU (+ second affine, + residual, + accumulate) as an implicit GEMM on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Replaces the nn.Conv2d -> nn.BatchNorm2d -> nn.ReLU chains of networks/surface_normal.py:10-145 and
// networks/depth_completion.py:16-147 (on the reference: one cuDNN/ATen call per layer, 462 convs + 458 BNs per frame).
//
//   M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, activations NHWC fp32, weights packed [Cout][KH][KW][Cin].
//   Workgroup tile BM x BN, K-step 32 (one (kh,kw) tap, 32 input channels = 128 contiguous bytes per pixel row).
//   Global -> registers (16 B/lane, issued one K-step ahead) -> LDS (rows padded to 36 floats) -> ds_read_b128
//   fragments.  Each lane's b128 holds 4 consecutive k of its row; MFMA t of a group consumes element t of the A and
//   of the B fragment, i.e. the k-order inside an 8-wide chunk is permuted identically for both operands.
//   Epilogue in registers: acc*scale1+shift1, relu, [*scale2+shift2, relu], [+residual, relu], [+= y], store NHWC
//   at a channel offset (concat-free skip connections).  Split-K writes fp32 partials and a finalize kernel applies
//   the same epilogue.  `groups` (blockIdx.z) runs the three ModifiedFPN pyramids in one launch.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;       // K-step (floats)
constexpr int LDS_LD = 36;   // padded LDS row (floats): 144 B keeps b128 reads of 16 consecutive rows conflict-free

struct ConvArgs {
    const float* x; const float* w; float* y;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    const float* residual; float* ws;
    int B, H, W, Cin, ldx, Ho, Wo, Cout, ldy, ldr, KH, KW, stride, pad, flags, groups;
    long long x_gs, w_gs, y_gs, r_gs, p_gs;
    int M, K, ksteps, splitk, tiles_m, tiles_n;
};

__device__ inline float epilogue(float v, int n, size_t off_r, size_t off_y, const ConvArgs& a, const float* s1, const float* b1,
                                 const float* s2, const float* b2, const float* res, const float* y) {
    v = v * s1[n] + b1[n];
    if (a.flags & VIDC_RELU1) v = fmaxf(v, 0.f);
    if (a.flags & VIDC_AFFINE2) {
        v = v * s2[n] + b2[n];
        if (a.flags & VIDC_RELU2) v = fmaxf(v, 0.f);
    }
    if (a.flags & VIDC_RESIDUAL) {
        v += res[off_r];
        if (a.flags & VIDC_RELU3) v = fmaxf(v, 0.f);
    }
    if (a.flags & VIDC_ACCUM) v += y[off_y];
    return v;
}

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN))
conv_igemm_f32(const ConvArgs a) {
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_ITERS = (BM * 8 + NT - 1) / NT;   // float4 loads per thread per K-step
    constexpr int B_ITERS = (BN * 8 + NT - 1) / NT;
    constexpr int ROWS_PER_PASS = NT / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;             // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int g = blockIdx.z;
    const int tile_n = blockIdx.x / a.tiles_m;      // consecutive blocks walk M: they share one weight tile (L2 reuse)
    const int tile_m = blockIdx.x - tile_n * a.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kz = blockIdx.y;
    const int ks_begin = (int)(((long long)a.ksteps * kz) / a.splitk);
    const int ks_end = (int)(((long long)a.ksteps * (kz + 1)) / a.splitk);

    const float* __restrict__ xg = a.x + g * a.x_gs;
    const float* __restrict__ wg = a.w + g * a.w_gs;

    // ---- per-thread load coordinates -----------------------------------------------------------------
    const int lrow = tid >> 3, c4 = (tid & 7) * 4;
    const float* a_ptr[A_ITERS];
    int a_iy0[A_ITERS], a_ix0[A_ITERS];
    bool a_ok[A_ITERS];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < A_ITERS; ++j) {
        int r = lrow + j * ROWS_PER_PASS;
        int m = m0 + r;
        bool ok = (r < BM) && (m < a.M);
        int mm = ok ? m : 0;
        int b = mm / HoWo, rem = mm - b * HoWo;
        int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        a_iy0[j] = oy * a.stride - a.pad;
        a_ix0[j] = ox * a.stride - a.pad;
        a_ok[j] = ok;
        a_ptr[j] = xg + ((long long)(b * a.H + a_iy0[j]) * a.W + a_ix0[j]) * a.ldx + c4;
    }
    const float* b_ptr[B_ITERS];
    bool b_ok[B_ITERS];
#pragma unroll
    for (int j = 0; j < B_ITERS; ++j) {
        int r = lrow + j * ROWS_PER_PASS;
        int n = n0 + r;
        b_ok[j] = (r < BN) && (n < a.Cout);
        b_ptr[j] = wg + (long long)(b_ok[j] ? n : 0) * a.K + c4;
    }

    // K-step -> (kh, kw, channel chunk), advanced incrementally
    const int cpt = a.Cin / BK;
    int tap = ks_begin / cpt, cc = ks_begin - tap * cpt;
    int kh = tap / a.KW, kw = tap - kh * a.KW;

    float4 ra[A_ITERS], rb[B_ITERS];
    auto load_tile = [&](int ks) {
        const long long tap_off = ((long long)kh * a.W + kw) * a.ldx + cc * BK;
#pragma unroll
        for (int j = 0; j < A_ITERS; ++j) {
            int iy = a_iy0[j] + kh, ix = a_ix0[j] + kw;
            bool ok = a_ok[j] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            ra[j] = ok ? *reinterpret_cast<const float4*>(a_ptr[j] + tap_off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j)
            rb[j] = b_ok[j] ? *reinterpret_cast<const float4*>(b_ptr[j] + (long long)ks * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (++cc == cpt) { cc = 0; if (++kw == a.KW) { kw = 0; ++kh; } }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_ITERS; ++j) {
            int r = lrow + j * ROWS_PER_PASS;
            if (A_ITERS * ROWS_PER_PASS == BM || r < BM)
                *reinterpret_cast<float4*>(&As[(buf * BM + r) * LDS_LD + c4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j) {
            int r = lrow + j * ROWS_PER_PASS;
            if (B_ITERS * ROWS_PER_PASS == BN || r < BN)
                *reinterpret_cast<float4*>(&Bs[(buf * BN + r) * LDS_LD + c4]) = rb[j];
        }
    };

    // ---- main loop -----------------------------------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int buf = 0;
    if (ks_begin < ks_end) {
        load_tile(ks_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const bool more = ks + 1 < ks_end;
        if (more) load_tile(ks + 1);     // global loads in flight under the MFMAs below
        const float* Ab = &As[(buf * BM + wm * WM + li) * LDS_LD + lh * 4];
        const float* Bb = &Bs[(buf * BN + wn * WN + li) * LDS_LD + lh * 4];
#pragma unroll
        for (int sub = 0; sub < BK / 8; ++sub) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + sub * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + sub * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -----------
    const float* s1 = a.scale1 + g * a.p_gs;
    const float* b1 = a.shift1 + g * a.p_gs;
    const float* s2 = a.scale2 ? a.scale2 + g * a.p_gs : nullptr;
    const float* b2 = a.shift2 ? a.shift2 + g * a.p_gs : nullptr;
    const float* res = a.residual ? a.residual + g * a.r_gs : nullptr;
    float* yg = a.y + g * a.y_gs;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + li;
            if (n >= a.Cout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                if (a.splitk > 1) {
                    a.ws[((size_t)(kz * a.groups + g) * a.M + m) * a.Cout + n] = acc[i][j][r];
                } else {
                    size_t oy = (size_t)m * a.ldy + n;
                    yg[oy] = epilogue(acc[i][j][r], n, (size_t)m * a.ldr + n, oy, a, s1, b1, s2, b2, res, yg);
                }
            }
        }
}

// Sums the split-K partials and applies the fused epilogue; one thread per 4 output channels.
__global__ void __launch_bounds__(256) conv_splitk_finalize(const ConvArgs a) {
    const int g = blockIdx.z;
    const int q = a.Cout / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.M * q) return;
    const int m = (int)(idx / q), n = (int)(idx - (long long)m * q) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < a.splitk; ++z) {
        float4 v = *reinterpret_cast<const float4*>(&a.ws[((size_t)(z * a.groups + g) * a.M + m) * a.Cout + n]);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float* s1 = a.scale1 + g * a.p_gs;
    const float* b1 = a.shift1 + g * a.p_gs;
    const float* s2 = a.scale2 ? a.scale2 + g * a.p_gs : nullptr;
    const float* b2 = a.shift2 ? a.shift2 + g * a.p_gs : nullptr;
    const float* res = a.residual ? a.residual + g * a.r_gs : nullptr;
    float* yg = a.y + g * a.y_gs;
    float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        size_t oy = (size_t)m * a.ldy + n + t;
        yg[oy] = epilogue(v[t], n + t, (size_t)m * a.ldr + n + t, oy, a, s1, b1, s2, b2, res, yg);
    }
}

__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                          int KH, int KW) {
    // wp[o][kh][kw][c] = w[o][c][kh][kw]
    const long long total = (long long)Cout * Cin * KH * KW;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int c = (int)(idx % Cin);
    long long t = idx / Cin;
    int kw = (int)(t % KW); t /= KW;
    int kh = (int)(t % KH);
    int o = (int)(t / KH);
    wp[idx] = w[(((long long)o * Cin + c) * KH + kh) * KW + kw];
}

struct TileInfo { int bm, bn, threads; };
constexpr TileInfo kTiles[VIDC_TILE_COUNT] = {{0, 0, 0}, {128, 128, 256}, {128, 64, 256}, {64, 128, 256}, {64, 64, 256},
                                              {32, 128, 256}, {32, 64, 128}};

template <int BM, int BN, int WM, int WN>
int launch_tile(const ConvArgs& a, hipStream_t st) {
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WM, WN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.splitk, a.groups);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN>), grid, dim3(NT), lds, st, a);
    VIDC_CHECK_LAUNCH("conv_igemm_f32");
    return VIDC_OK;
}

int validate(const vidc_conv_desc* d) {
    VIDC_REQUIRE(d, VIDC_ERR_NULL, "conv: null descriptor");
    VIDC_REQUIRE(d->x && d->w && d->y && d->scale1 && d->shift1, VIDC_ERR_NULL, "conv: null tensor pointer");
    VIDC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, VIDC_ERR_SHAPE, "conv: bad spatial dims");
    VIDC_REQUIRE(d->Cin > 0 && d->Cin % BK == 0, VIDC_ERR_SHAPE, "conv: Cin=%d must be a multiple of %d", d->Cin, BK);
    VIDC_REQUIRE(d->Cout > 0 && d->Cout % 4 == 0, VIDC_ERR_SHAPE, "conv: Cout=%d must be a multiple of 4", d->Cout);
    VIDC_REQUIRE(d->ldx >= d->Cin && d->ldy >= d->Cout && d->ldx % 4 == 0, VIDC_ERR_SHAPE, "conv: bad channel strides");
    VIDC_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->stride >= 1 && d->pad >= 0, VIDC_ERR_SHAPE, "conv: bad kernel geometry");
    VIDC_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                 VIDC_ERR_SHAPE, "conv: Ho/Wo inconsistent with H/W, kernel, stride, pad");
    VIDC_REQUIRE(d->groups >= 1 && d->splitk >= 1, VIDC_ERR_SHAPE, "conv: groups/splitk must be >= 1");
    VIDC_REQUIRE(!(d->flags & VIDC_AFFINE2) || (d->scale2 && d->shift2), VIDC_ERR_NULL, "conv: AFFINE2 without scale2/shift2");
    VIDC_REQUIRE(!(d->flags & VIDC_RESIDUAL) || (d->residual && d->ldr >= d->Cout), VIDC_ERR_NULL, "conv: RESIDUAL without tensor");
    VIDC_REQUIRE(d->tile >= 0 && d->tile < VIDC_TILE_COUNT, VIDC_ERR_SHAPE, "conv: unknown tile id %d", d->tile);
    VIDC_REQUIRE(d->splitk == 1 || d->workspace, VIDC_ERR_NULL, "conv: split-K needs a workspace");
    VIDC_REQUIRE((long long)d->B * d->Ho * d->Wo < (1ll << 31), VIDC_ERR_SHAPE, "conv: M overflows int32");
    return VIDC_OK;
}

}  // namespace

extern "C" int vidc_conv2d_plan(vidc_conv_desc* d) {
    VIDC_REQUIRE(d, VIDC_ERR_NULL, "conv plan: null descriptor");
    // Cost model: every SIMD issues one 32x32x2 MFMA per 64 clk; a workgroup keeps the 4 SIMDs of one CU busy, so the
    // time of a launch ~ ceil(workgroups / 256 CUs) * (MFMAs per wave) + fixed prologue/epilogue; split-K adds a pass
    // over the partials.  Pick the (tile, splitk) minimising that.
    const long long M = (long long)d->B * d->Ho * d->Wo;
    const int ksteps = d->KH * d->KW * d->Cin / BK;
    int n_cu = 256;
    double best = 1e30;
    int best_tile = VIDC_TILE_64x64, best_sk = 1;
    for (int t = 1; t < VIDC_TILE_COUNT; ++t) {
        const TileInfo ti = kTiles[t];
        if (ti.bn > d->Cout && ti.bn != 64) continue;
        const long long tm = (M + ti.bm - 1) / ti.bm, tn = (d->Cout + ti.bn - 1) / ti.bn;
        const int waves = ti.threads / 64;
        const double mfma_per_wave_kstep = (double)(ti.bm / 32) * (ti.bn / 32) / waves * 16.0;
        for (int sk = 1; sk <= 16; sk *= 2) {
            if (sk > 1 && ksteps / sk < 8) break;
            const long long wgs = tm * tn * sk * d->groups;
            // a CU hosts 4 waves per "slot"; smaller workgroups pack (256/threads) per slot
            const double slots = (double)wgs * ti.threads / 256.0;
            const double rounds = ceil(slots / n_cu);
            double cyc = rounds * ((double)((ksteps + sk - 1) / sk) * mfma_per_wave_kstep * 64.0 + 3000.0);
            if (sk > 1) cyc += 4000.0 + (double)M * d->Cout * d->groups * (sk + 1) * 4.0 / 2500.0;   // ~6 TB/s @2.4 GHz
            if (cyc < best) { best = cyc; best_tile = t; best_sk = sk; }
        }
    }
    d->tile = best_tile;
    d->splitk = best_sk;
    return VIDC_OK;
}

extern "C" size_t vidc_conv2d_workspace_bytes(const vidc_conv_desc* d) {
    if (!d || d->splitk <= 1) return 0;
    return (size_t)d->splitk * d->groups * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
}

extern "C" int vidc_conv2d_bn_act(const vidc_conv_desc* d, vidc_stream_t stream) {
    int rc = validate(d);
    if (rc != VIDC_OK) return rc;
    vidc_conv_desc dd = *d;
    if (dd.tile == VIDC_TILE_AUTO) {
        int sk = dd.splitk;
        vidc_conv2d_plan(&dd);
        if (!dd.workspace) dd.splitk = 1; else if (sk > 1) dd.splitk = sk;
    }
    ConvArgs a;
    a.x = dd.x; a.w = dd.w; a.y = dd.y; a.scale1 = dd.scale1; a.shift1 = dd.shift1; a.scale2 = dd.scale2; a.shift2 = dd.shift2;
    a.residual = dd.residual; a.ws = dd.workspace;
    a.B = dd.B; a.H = dd.H; a.W = dd.W; a.Cin = dd.Cin; a.ldx = dd.ldx; a.Ho = dd.Ho; a.Wo = dd.Wo; a.Cout = dd.Cout;
    a.ldy = dd.ldy; a.ldr = dd.ldr; a.KH = dd.KH; a.KW = dd.KW; a.stride = dd.stride; a.pad = dd.pad; a.flags = dd.flags;
    a.groups = dd.groups; a.x_gs = dd.x_gs; a.w_gs = dd.w_gs; a.y_gs = dd.y_gs; a.r_gs = dd.r_gs; a.p_gs = dd.p_gs;
    a.M = dd.B * dd.Ho * dd.Wo; a.K = dd.KH * dd.KW * dd.Cin; a.ksteps = a.K / BK;
    a.splitk = dd.splitk > a.ksteps ? a.ksteps : dd.splitk;
    const TileInfo ti = kTiles[dd.tile];
    a.tiles_m = (a.M + ti.bm - 1) / ti.bm;
    a.tiles_n = (a.Cout + ti.bn - 1) / ti.bn;
    hipStream_t st = vidc::as_stream(stream);
    switch (dd.tile) {
        case VIDC_TILE_128x128: rc = launch_tile<128, 128, 64, 64>(a, st); break;
        case VIDC_TILE_128x64:  rc = launch_tile<128, 64, 64, 32>(a, st); break;
        case VIDC_TILE_64x128:  rc = launch_tile<64, 128, 32, 64>(a, st); break;
        case VIDC_TILE_64x64:   rc = launch_tile<64, 64, 32, 32>(a, st); break;
        case VIDC_TILE_32x128:  rc = launch_tile<32, 128, 32, 32>(a, st); break;
        case VIDC_TILE_32x64:   rc = launch_tile<32, 64, 32, 32>(a, st); break;
        default: VIDC_REQUIRE(false, VIDC_ERR_SHAPE, "conv: bad tile");
    }
    if (rc != VIDC_OK) return rc;
    if (a.splitk > 1) {
        long long n4 = (long long)a.M * (a.Cout / 4);
        dim3 grid((unsigned)((n4 + 255) / 256), 1, a.groups);
        hipLaunchKernelGGL(conv_splitk_finalize, grid, dim3(256), 0, st, a);
        VIDC_CHECK_LAUNCH("conv_splitk_finalize");
    }
    return VIDC_OK;
}

extern "C" int vidc_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW,
                                     vidc_stream_t stream) {
    VIDC_REQUIRE(w_oihw && w_packed, VIDC_ERR_NULL, "vidc_pack_conv_weight: null pointer");
    VIDC_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, VIDC_ERR_SHAPE, "vidc_pack_conv_weight: bad shape");
    long long total = (long long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), w_oihw,
                       w_packed, Cout, Cin, KH, KW);
    VIDC_CHECK_LAUNCH("pack_weight_kernel");
    return VIDC_OK;
}
