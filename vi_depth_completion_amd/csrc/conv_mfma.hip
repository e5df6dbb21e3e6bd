// Fused conv + BN + ReLU (+ second affine, + residual, + accumulate) as an implicit GEMM on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Replaces the nn.Conv2d -> nn.BatchNorm2d -> nn.ReLU chains of networks/surface_normal.py:10-145 and
// networks/depth_completion.py:16-147 (on the reference: one cuDNN/ATen call per layer, 462 convs + 458 BNs per frame).
//
//   M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, activations NHWC fp32, weights packed [Cout][Cin/32][KH][KW][32].
//   Workgroup tile BM x BN, K unit 32 floats (one (kh,kw) tap, 32 input channels = 128 contiguous bytes per pixel row).
//   Global -> LDS by DMA (global_load_lds, 16 B/lane) into an NS-deep ring with counted vmcnt waits and raw barriers,
//   so NS-1 stages of weights/activations are in flight per workgroup (weights are always HBM-cold: 1.5 GB per frame).
//   LDS -> ds_read_b128 fragments.  Each lane's b128 holds 4 consecutive k of its row; MFMA t of a group consumes
//   element t of the A and of the B fragment, i.e. the k-order inside an 8-wide chunk is permuted identically for both.
//   Epilogue in registers: acc*scale1+shift1, relu, [*scale2+shift2, relu], [+residual, relu], [+= y], store NHWC
//   at a channel offset (concat-free skip connections).  Split-K: every k-slice workgroup writes its fp32 partial tile and
//   the last one to arrive (ticket counter per tile) sums them in slice order and runs the same epilogue -- no second launch.  `groups` (blockIdx.z) runs the three ModifiedFPN pyramids in one launch.
#include "common.h"
#include <type_traits>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;       // K-step (floats)

struct ConvArgs {
    const float* x; const float* w; float* y;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    const float* residual; float* ws; unsigned short* y_split;
    int B, H, W, Cin, ldx, Ho, Wo, Cout, ldy, ldr, KH, KW, stride, pad, flags, groups, dil;
    long long x_gs, w_gs, y_gs, r_gs, p_gs;
    int M, K, ksteps, splitk, tiles_m, tiles_n;
    unsigned dv_tiles_m[3], dv_splitk[3], dv_tiles_n[3], dv_ntaps[3], dv_kw[3];   // {multiplier, shift, d == 1 mask} of fast_div()
};

// Division of a wave-uniform n < 2^31 by a launch constant d without the ~40-instruction emulated divide: the host stores
// mul = ceil(2^(31+l) / d), l = ceil(log2 d); n / d == umulhi(n, mul) >> (l - 1), exact for every n < 2^31 (d >= 2).
// d == 1 is encoded as mul = 0 plus an all-ones mask that passes n through, so the device side is branch-free (a branch
// would also split the prologue into basic blocks, each with its own kernel-argument s_load + wait).
__host__ inline void fast_div_init(unsigned d, unsigned out[3]) {
    if (d <= 1) { out[0] = 0; out[1] = 0; out[2] = 0xFFFFFFFFu; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long num = 1ull << (31 + l);
    out[0] = (unsigned)((num + d - 1) / d);
    out[1] = l - 1;
    out[2] = 0;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const unsigned dv[3]) {
    return (__umulhi(n, dv[0]) >> dv[1]) + (n & dv[2]);
}

using vidc::bf16_rne;
using vidc::split_bf16;
using vidc::store_split;

// Zero source for LDS-DMA lanes whose row is padding (conv halo, M tail, Cout tail, K tail).
__device__ float g_zero_chunk[64] = {0};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n in [0, MAXN]; the instruction needs an immediate.  Tested from MAXN
// downwards: the steady-state value is MAXN, so a main-loop iteration pays one scalar compare.  Waiting for a smaller
// count than necessary is always safe.
template <int MAXN, int N = MAXN>
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    if constexpr (N > 0) {
        if (n >= N) { wait_vmcnt<N>(); return; }
        wait_vmcnt_dyn<MAXN, N - 1>(n);
    } else {
        wait_vmcnt<0>();
    }
}

// Stage s has landed when only DMA groups younger than its own are outstanding.  Issue order: prologue B_0..B_{PRO-1},
// A_0..A_{PRO-1} (PRO = NS-1; always issued, zero-sourced past the K range), then one (A, B) pair per main-loop iteration
// s = 0 .. nst-NS; the last NS-1 iterations issue nothing.
template <int NS, int A_J, int LPS>
__device__ __forceinline__ void wait_stage(int s, int nst) {
    constexpr int PRO = NS - 1;
    const int n_main = nst > PRO ? nst - PRO : 0;
    const int loop_issued = s < n_main ? s : n_main;                    // (A, B) pairs issued by the loop before this wait
    int younger;
    if (s < PRO) younger = (PRO - 1 - s) * A_J + loop_issued * LPS;     // its A group is in the prologue's A run
    else         younger = (loop_issued - (s - PRO) - 1) * LPS;         // stage s was issued by loop iteration s-PRO
    wait_vmcnt_dyn<(NS - 2) * LPS>(younger);      // younger <= (NS-2)*LPS always (A_J <= LPS)
}
// The same count for a wait that is not taken in the iteration that consumes the stage: stage q has landed, seen from loop iteration `it`
// (which has issued min(it, n_main) (A, B) pairs so far).  SPEC 2 loaders wait for stage it + 1 before the barrier of iteration it.
template <int NS, int A_J, int LPS>
__device__ __forceinline__ void wait_landed(int q, int it, int nst) {
    constexpr int PRO = NS - 1;
    const int n_main = nst > PRO ? nst - PRO : 0;
    const int loop_issued = it < n_main ? it : n_main;
    int younger = q < PRO ? (PRO - 1 - q) * A_J + loop_issued * LPS : (loop_issued - (q - PRO) - 1) * LPS;
    if (younger < 0) younger = 0;
    wait_vmcnt_dyn<(NS - 2) * LPS>(younger);      // (a count above the template bound waits for more than necessary: safe)
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// Workgroup = WMW x WNW x WKW waves.  The (WMW x WNW) waves of one k-slice tile the BM x BN output; the WKW k-slices
// split every pipeline stage's K range (32 floats each) and are summed through LDS at the end (deterministic order).
// NS-deep LDS ring filled by global_load_lds (16 B per lane, 1 KiB = 8 rows x 128 B per wave-instruction); the 16-byte
// chunks of a row are XOR-swizzled by ((row >> 1) & 7) through the per-lane SOURCE address, so the DMA destination stays lane-linear
// and ds_read_b128 fragments are bank-conflict free: the LDS is 256 B = two 128-byte rows wide, a ds_read_b128 is served in four
// 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), and within every group the eight even rows have eight different
// (row >> 1) & 7 and so do the eight odd ones -> 16 different 16-byte slots.  (The first version keyed on row & 7: rows r and r + 8 of
// a group then shared a slot, every fragment read was 2-way conflicted and the LDS delivered 128 instead of 256 B/clk.)
// PREC 0: fp32 operands, v_mfma_f32_32x32x2_f32.  PREC 1 ("bf16x3"): both operands arrive pre-split as bf16 hi + lo
// (each 32-channel K unit is stored as [32 x hi | 32 x lo] = the same 128 bytes per row as fp32, so addressing, DMA and
// swizzle are identical) and every K unit is 3 x 2 v_mfma_f32_32x32x16_bf16: lo*hi + hi*lo + hi*hi, fp32 accumulate.
// Dropping lo*lo leaves ~2^-16 relative error per product: depth RMSE 1.4e-5 vs fp32 over the whole path (bar: 1e-3).
// PREC 2 ("bf16", the training mode BASELINE configs[4] names): plain bf16 operands, fp32 accumulate.  A 128-byte unit of a row holds
// 64 bf16 channels, so the caller describes the tensors in units of two channels (Cin, ldx = bf16 channels / 2, weights packed per 64
// channels) and everything up to the fragment reads is unchanged; a K unit is 2 x 2 v_mfma_f32_32x32x16_bf16.
// SPEC 1 ("loader waves"): the workgroup has NW extra waves that do nothing but the LDS-DMA of the NW compute waves (loader l
// issues exactly what compute wave l would) and the compute waves issue no DMA at all.  A wave can issue one 1 KiB DMA per ~64 clk
// and stalls in-order behind it, so in the small-tile kernels (6 DMAs per wave per 192 clk of MFMA) the DMA issue used to sit on
// the compute waves' critical path (in-kernel stamps, M=320 layer-3 shapes: 853 clk per stage, 521 clk without the refills,
// 829 clk without the MFMAs).  One loader and one compute wave share each SIMD; the per-stage s_barrier is the only hand-off:
// a loader passes it after ITS loads of stage s have landed (its own vmcnt), a compute wave after it has read stage s-1.
// SPEC 2 ("pipelined fragment reads", round 3; bf16 modes only, fp32 falls back to SPEC 1): loader waves as in SPEC 1, and the compute
// waves never wait for LDS in front of an idle matrix pipe.  With one compute wave per SIMD (all a batch-1 layer has: 960 wave tiles of
// 64x64 for 1024 SIMDs) the SPEC 1 loop -- barrier, 16 ds_read_b128, wait, 12 MFMAs, wait, 12 MFMAs -- runs the pipe at 67 %
// (tools/ubench/mfma_loop.hip: 1143 clk per 768 clk of MFMA; priorities and a second wave per SIMD do not change that).  Here every
// fragment read sits behind an MFMA: the reads of a stage's second k-half are interleaved with the MFMAs of its first half, and the reads
// of the NEXT stage's first half with the MFMAs of the second half, ACROSS the stage barrier (849 clk per stage in the micro-benchmark,
// 90 %).  For that the barrier of iteration s must also guarantee that stage s + 1 has landed: the loaders wait one stage further
// ahead (wait_landed(s + 1)), which costs one stage of DMA look-ahead -- the ring is one slot deeper (NS = 4 for the 128x128 tile).
// Per accumulator the MFMA order is unchanged (lo*hi, hi*lo, hi*hi of k-half 0, then of k-half 1), so results are bit-identical to SPEC 1.
template <int BM, int BN, int WMW, int WNW, int WKW, int NS, int PREC, int SPEC>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int g, const int tile_m, const int tile_n, const int kz, float* smem) {
    constexpr int NW = WMW * WNW * WKW, WPK = WMW * WNW;
    constexpr int TM = BM / (32 * WMW), TN = BN / (32 * WNW);
    constexpr int A_J = (BM / 8) / WPK, B_J = (BN / 8) / WPK;      // DMA instructions per wave per stage
    constexpr int LPS = A_J + B_J;
    constexpr int STAGE = (BM + BN) * BK * WKW;                    // floats per ring slot
    static_assert((BM / 8) % WPK == 0 && (BN / 8) % WPK == 0, "tile rows must split evenly over the waves");
    static_assert(NS >= 2 && (NS - 2) * LPS <= 63, "vmcnt is a 6-bit counter");
    static_assert((WKW - 1) * WPK * TM * TN * 1024 <= NS * STAGE, "K-reduction scratch must fit in the ring");
#ifdef VIDC_CONV_TIMING
    // debug build only: per-workgroup phase stamps (shader clock + 100 MHz wall clock) into a.ws
    long long* dbg = reinterpret_cast<long long*>(a.ws) + (size_t)blockIdx.x * 16;
    const long long t_begin = __builtin_readcyclecounter();
#define VIDC_STAMP(k) do { if (threadIdx.x == 0) dbg[k] = __builtin_readcyclecounter() - t_begin; } while (0)
    if (threadIdx.x == 0) dbg[6] = (long long)__builtin_amdgcn_s_memrealtime();
#else
#define VIDC_STAMP(k) do {} while (0)
#endif

    const int tid = threadIdx.x;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int SPEC_EFF = (SPEC == 2 && PREC == 0) ? 1 : SPEC;      // the pipelined loop exists for the 16-k bf16 MFMAs only
    const bool is_loader = SPEC && wave_all >= NW;              // wave-uniform
    const bool loads = !SPEC || is_loader;                      // this wave issues DMA
    const int wave = is_loader ? wave_all - NW : wave_all;      // role-local index: loader l feeds what compute wave l would load
    const int kq = wave / WPK, wq = wave - kq * WPK;
    const int wm = wq / WNW, wn = wq - wm * WNW;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    VIDC_STAMP(8);      // kernel arguments arrived, work item decoded
    const int units = a.ksteps;                                       // K in units of 32 floats
    const int stages_total = (units + WKW - 1) / WKW;
    const int st_begin = (int)fast_div((unsigned)(stages_total * kz), a.dv_splitk);          // stages_total * splitk < 2^31
    const int st_end = (int)fast_div((unsigned)(stages_total * (kz + 1)), a.dv_splitk);
    const int nst = st_end - st_begin;

    // ---- per-lane DMA source coordinates -----------------------------------------------------------------
    // Buffer addressing: one descriptor per operand (SGPRs) + a 32-bit byte offset per lane.  A lane whose row is
    // padding (conv halo, M tail, Cout tail, K tail) gets an offset beyond num_records and the hardware returns zeros,
    // so there is no per-lane pointer select and no 64-bit address arithmetic in the loop.
    constexpr unsigned OOB = 0x80000000u;
    // (descriptor inputs go through readfirstlane so that hipcc can prove them wave-uniform; otherwise it wraps every
    //  buffer op in a waterfall loop -- guide T20)
    auto uniform_ptr = [](const float* p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    };
    // (VIDC_X_PLANAR_GROUPS: every group's input is a plane of its own with B*H*W rows of ldx values, x_gs apart -- the training step's grouped
    //  weight-gradient GEMMs; else the groups are channel slices of shared rows and a later group sees fewer bytes behind its base)
    const int x_bytes = __builtin_amdgcn_readfirstlane((int)(((long long)a.B * a.H * a.W * a.ldx - ((a.flags & VIDC_X_PLANAR_GROUPS) ? 0 : g * a.x_gs)) * 4));
    const int w_bytes = __builtin_amdgcn_readfirstlane((int)((long long)a.Cout * a.K * 4));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.x + g * a.x_gs), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.w + g * a.w_gs), 0, w_bytes, 0x00020000);
    const int lrow = lane >> 3;
    // swizzled 16-byte chunk (in floats) this lane fetches for the j-th DMA of a stage: tile row = ... + (j * WPK + wq) * 8 + lrow, all
    // other terms multiples of 32, so (row >> 1) & 7 = ((j * WPK + wq) & 1) * 4 + (lrow >> 1)
    auto csw_of = [&](int j) { return ((lane & 7) ^ ((((j * WPK + wq) & 1) << 2) | (lrow >> 1))) * 4; };
    // ---- weights first: their DMAs need only (n, K unit) and they are the HBM-cold operand, so the prologue stages of B are in
    //      flight while the activation rows are decoded below ----------------------------------------------------------------
    unsigned b_off[B_J];
#pragma unroll
    for (int j = 0; j < B_J; ++j) {
        const int n = n0 + (j * WPK + wq) * 8 + lrow;
        b_off[j] = n < a.Cout ? (unsigned)((n * a.K + csw_of(j)) * 4) : OOB;
    }
    // K order: unit u = cu * (KH*KW) + tap -- channel unit major, tap minor -- so the KH*KW taps of one 32-channel unit run
    // back to back and the shifted re-reads of the same pixels hit L2 (tap-major order re-read every pixel KH*KW times with
    // ~10 MB of other traffic per XCD in between: 23x over-fetch measured on the 3x3, K=6912 layer).  This wave's unit
    // advances by WKW per stage.
    const int ntaps = a.KH * a.KW;
    int unit = st_begin * WKW + kq;
    int cc = (int)fast_div((unsigned)unit, a.dv_ntaps);
    int tap = unit - cc * ntaps;

    const int unit_end = min(units, st_end * WKW);      // past this workgroup's K range every DMA lane fetches zeros
    constexpr int PRO = NS - 1;                      // stages issued before the main loop
    if (loads) {
        int ub = unit;
#pragma unroll
        for (int s = 0; s < PRO; ++s) {
            const unsigned uoff = ub < unit_end ? (unsigned)(ub * BK * 4) : OOB;
#pragma unroll
            for (int j = 0; j < B_J; ++j) {
                float* dst = smem + s * STAGE + (WKW * BM + kq * BN + (j * WPK + wq) * 8) * BK;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t*)dst, 16, (int)(b_off[j] + uoff), 0, 0, 0);
            }
            ub += WKW;
        }
    }

    VIDC_STAMP(9);      // weight prologue DMAs issued
    int a_off[A_J];                                      // byte offset of (pixel, tap (0,0), channel csw); may be negative
    unsigned a_taps[A_J];                                // bit t set: tap t = kh*KW+kw of this row reads a real pixel
    const int HoWo = a.Ho * a.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;
    const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;    // input pixel index == output pixel index
#pragma unroll
    for (int j = 0; j < A_J; ++j) { a_off[j] = 0; a_taps[j] = 0u; }
#pragma unroll
    for (int j = 0; j < A_J; ++j) {
        if (!loads) break;                               // compute waves of a SPEC kernel never issue DMA
        const int m = m0 + (j * WPK + wq) * 8 + lrow;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        if (pointwise) {
            a_taps[j] = ok ? 1u : 0u;
            a_off[j] = (mm * a.ldx + csw_of(j)) * 4;
            continue;
        }
        int b = (int)((float)mm * inv_howo);             // float reciprocal + fix-up: exact for M < 2^23
        int rem = mm - b * HoWo;
        if (rem < 0) { --b; rem += HoWo; } else if (rem >= HoWo) { ++b; rem -= HoWo; }
        int oy = (int)((float)rem * inv_wo);
        int ox = rem - oy * a.Wo;
        if (ox < 0) { --oy; ox += a.Wo; } else if (ox >= a.Wo) { ++oy; ox -= a.Wo; }
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        unsigned colm = 0, taps = 0;                     // KH + KW steps instead of KH * KW; KH, KW <= 3 (validate())
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) colm |= (tw < a.KW && (unsigned)(ix0 + tw * a.dil) < (unsigned)a.W) ? 1u << tw : 0u;
#pragma unroll
        for (int th = 0; th < 3; ++th) taps |= (th < a.KH && (unsigned)(iy0 + th * a.dil) < (unsigned)a.H) ? colm << (th * a.KW) : 0u;
        a_taps[j] = ok ? taps : 0u;
        a_off[j] = (((b * a.H + iy0) * a.W + ix0) * a.ldx + csw_of(j)) * 4;
    }
    VIDC_STAMP(10);     // activation rows decoded
    // One pipeline stage = A_J + B_J DMA instructions per wave, issued in two halves that the main loop places inside
    // groups of MFMAs (branch-free, so the scheduler can interleave them with the 64-clk MFMA issue slots).
    auto issue_a = [&](int slot) {
        float* sbase = smem + slot * STAGE;
        const unsigned tapbit = unit < unit_end ? 1u << tap : 0u;
        const int kh = (int)fast_div((unsigned)tap, a.dv_kw), kw = tap - kh * a.KW;
        const int tap_off = ((((kh * a.W + kw) * a.dil) * a.ldx) + cc * BK) * 4;      // dilated taps (ASPP, surface_normal_dorn.py:46-62)
#pragma unroll
        for (int j = 0; j < A_J; ++j) {
            const unsigned voff = (a_taps[j] & tapbit) ? (unsigned)(a_off[j] + tap_off) : OOB;
            float* dst = sbase + (kq * BM + (j * WPK + wq) * 8) * BK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_t*)dst, 16, (int)voff, 0, 0, 0);
        }
    };
    auto issue_b = [&](int slot) {
        float* sbase = smem + slot * STAGE;
        const unsigned uoff = unit < unit_end ? (unsigned)(unit * BK * 4) : OOB;   // valid + OOB stays >= 2^31
#pragma unroll
        for (int j = 0; j < B_J; ++j) {
            float* dst = sbase + (WKW * BM + kq * BN + (j * WPK + wq) * 8) * BK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t*)dst, 16, (int)(b_off[j] + uoff), 0, 0, 0);
        }
    };
    // K position after one stage (this wave's unit advances by WKW): branch-free, so it can sit between the MFMAs of the loop
    const int step_cc = (int)fast_div((unsigned)WKW, a.dv_ntaps), step_tap = WKW - step_cc * ntaps;
    auto advance = [&]() {
        unit += WKW;
        tap += step_tap;
        const int wrap = tap >= ntaps ? 1 : 0;
        tap -= wrap ? ntaps : 0;
        cc += step_cc + wrap;
    };

    // ---- epilogue operands, fetched now so their (cold) latency hides under the main loop --------------------------
    const int li = lane & 31, lh = lane >> 5;
    const float* res = a.residual ? a.residual + g * a.r_gs : nullptr;
    float* yg = a.y + g * a.y_gs;
    float e_s1[TN], e_b1[TN], e_s2[TN], e_b2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { e_s1[j] = 1.f; e_b1[j] = 0.f; e_s2[j] = 1.f; e_b2[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (is_loader) break;
        const int n = n0 + wn * 32 * TN + j * 32 + li;
        const bool nok = n < a.Cout;
        const int ni = (int)(g * a.p_gs) + (nok ? n : 0);
        e_s1[j] = a.scale1[ni]; e_b1[j] = a.shift1[ni];
        e_s2[j] = 1.f; e_b2[j] = 0.f;
        if (a.flags & VIDC_AFFINE2) { e_s2[j] = a.scale2[ni]; e_b2[j] = a.shift2[ni]; }
    }
    VIDC_STAMP(11);     // scale/shift loads issued
    float e_res[TM][TN][16];
    if ((a.flags & VIDC_RESIDUAL) && a.splitk == 1 && !is_loader) {    // uniform branch; indices clamped so every load is unconditional
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = min(n0 + wn * 32 * TN + j * 32 + li, a.Cout - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.M - 1);
                    e_res[i][j][r] = res[(size_t)m * a.ldr + n];
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) e_res[i][j][r] = 0.f;
    }

    // ---- main loop -----------------------------------------------------------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    VIDC_STAMP(0);      // setup done
    const int n_main = nst > NS - 1 ? nst - (NS - 1) : 0;
    if constexpr (SPEC) {
        if (is_loader) {
            // ---- loader wave: DMA only.  Same issue order as the unspecialised kernel (prologue B..., A..., then A, B per
            //      iteration), so wait_stage()'s vmcnt arithmetic holds unchanged.  The last NS-1 iterations issue nothing and
            //      wait for everything, so no DMA is outstanding when the wave ends.
#pragma unroll
            for (int s = 0; s < PRO; ++s) { issue_a(s); advance(); }
            int slot = 0;
            for (int s = 0; s < nst; ++s) {
                if constexpr (SPEC_EFF == 2) wait_landed<NS, A_J, LPS>(s + 1 < nst ? s + 1 : nst - 1, s, nst);      // ... and stage s + 1
                else wait_stage<NS, A_J, LPS>(s, nst);
                __builtin_amdgcn_s_barrier();     // stage s is in LDS (every loader waited for its pieces); stage s-1 has been read
                int fill = slot + NS - 1; if (fill >= NS) fill -= NS;
                if (s < n_main) { issue_a(fill); issue_b(fill); advance(); }
                if (++slot == NS) slot = 0;
            }
            return;                               // terminated waves do not take part in the barriers of the epilogue
        }
    } else {
#pragma unroll
        for (int s = 0; s < PRO; ++s) { issue_a(s); advance(); }     // the B halves of these stages are already in flight
    }
    VIDC_STAMP(1);      // prologue DMAs issued

    // Fragment reads are inline asm: hipcc cannot prove that a ds_read does not alias an in-flight LDS-DMA and would
    // otherwise drain the whole ring (s_waitcnt vmcnt(0)) before the first read of every stage.  Ordering is ours:
    // counted lgkmcnt + sched_barrier(0) before the MFMAs that consume the registers (guide §5.7 form iii, rule 18).
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) float*)smem);
    const unsigned a_base = lds0 + 4u * (unsigned)((kq * BM + wm * 32 * TM + li) * BK);
    const unsigned b_base = lds0 + 4u * (unsigned)((WKW * BM + kq * BN + wn * 32 * TN + li) * BK);
    const int sw = (li >> 1) & 7;       // = (tile row >> 1) & 7: the fragment rows start at multiples of 32
    unsigned coff[BK / 8];
#pragma unroll
    for (int sub = 0; sub < BK / 8; ++sub) coff[sub] = (unsigned)(((sub * 2 + lh) ^ sw) * 16);

    // One pipeline iteration.  `issue_tag` = true_type: refill the ring (branch-free, so the DMA issue can be scheduled
    // between MFMAs); false_type: the last NS-1 iterations, which consume what is already in flight and issue nothing --
    // so no DMA is outstanding when the loop ends and the epilogue does not have to drain any.
    int slot = 0;
    f32x4 fa[2][TM], fb[2][TN];            // fp32 fragments, double-buffered from one 8-k sub-step to the next
    auto iteration = [&](int s, auto issue_tag) {
        // DMA issue order: prologue B_0..B_{PRO-1}, A_0..A_{PRO-1}, then per iteration A, B.  Stage s < PRO has landed when
        // only the (PRO-1-s) younger prologue A groups and the s stages issued by the loop remain; from s = PRO on, when at
        // most NS-2 whole stages remain.
        if constexpr (!SPEC) wait_stage<NS, A_J, LPS>(s, nst);      // SPEC: the loader waves wait for their DMA before this barrier
        __builtin_amdgcn_s_barrier();     // every wave's pieces of stage s are in LDS; everyone finished stage s-1
        if (s == 0) VIDC_STAMP(2);      // first stage landed
        int fill = slot + NS - 1; if (fill >= NS) fill -= NS;     // the slot read in iteration s-1: free since the barrier
        const unsigned Ab = a_base + (unsigned)(slot * STAGE * 4);
        const unsigned Bb = b_base + (unsigned)(slot * STAGE * 4);
        if constexpr (PREC != 0) {
            // logical 16-byte chunks of a row's unit: 0..3 = hi (k 0-7, 8-15, 16-23, 24-31), 4..7 = lo; MFMA t covers
            // k 16t..16t+15 with lanes 0-31 supplying the first and lanes 32-63 the second 8 k.
            f32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned ch = (unsigned)(((2 * t + lh) ^ sw) * 16), cl = (unsigned)(((4 + 2 * t + lh) ^ sw) * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[t][i] = lds_read_b128<0>(Ab + ch + i * 32 * BK * 4);
                    al[t][i] = lds_read_b128<0>(Ab + cl + i * 32 * BK * 4);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[t][j] = lds_read_b128<0>(Bb + ch + j * 32 * BK * 4);
                    bl[t][j] = lds_read_b128<0>(Bb + cl + j * 32 * BK * 4);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 0) wait_lgkmcnt<2 * (TM + TN)>(); else wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
#ifndef VIDC_DBG_SKIP_DMA
                if constexpr (decltype(issue_tag)::value && !SPEC) { if (t == 0) issue_a(fill); else { issue_b(fill); advance(); } }   // among the MFMAs
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[t][i]), xl = __builtin_bit_cast(bf16x8, al[t][i]);
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[t][j]), wl = __builtin_bit_cast(bf16x8, bl[t][j]);
#ifdef VIDC_DBG_SKIP_MFMA
                        acc[i][j][0] += __builtin_bit_cast(f32x4, xl).x + __builtin_bit_cast(f32x4, wh).x + __builtin_bit_cast(f32x4, xh).x + __builtin_bit_cast(f32x4, wl).x;
#else
                        if constexpr (PREC == 2) {      // plain bf16: chunks 4..7 are channels 32..63 of the 64-channel unit
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wl, acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc[i][j], 0, 0, 0);
                        }
#endif
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
    #pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = lds_read_b128<0>(Ab + coff[0] + i * 32 * BK * 4);
    #pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = lds_read_b128<0>(Bb + coff[0] + j * 32 * BK * 4);
    #pragma unroll
            for (int sub = 0; sub < BK / 8; ++sub) {
                const int cur = sub & 1, nxt = cur ^ 1;
                if (sub + 1 < BK / 8) {
    #pragma unroll
                    for (int i = 0; i < TM; ++i) fa[nxt][i] = lds_read_b128<0>(Ab + coff[sub + 1] + i * 32 * BK * 4);
    #pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = lds_read_b128<0>(Bb + coff[sub + 1] + j * 32 * BK * 4);
                    wait_lgkmcnt<TM + TN>();      // the reads of `cur` are complete (LDS returns in order)
                } else {
                    wait_lgkmcnt<0>();
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (decltype(issue_tag)::value && !SPEC) {
                    if (sub == 0) issue_a(fill);     // scheduled among the MFMAs below
                    if (sub == 1) { issue_b(fill); advance(); }
                }
    #pragma unroll
                for (int i = 0; i < TM; ++i)
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fb[cur][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fb[cur][j].w, acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++slot == NS) slot = 0;
    };
    if constexpr (SPEC_EFF == 2) {
        static_assert(PREC != 0 && NS >= 3, "pipelined fragment reads: bf16 modes, ring of at least 3");
        constexpr int NP = PREC == 1 ? 3 : 2;                      // MFMA passes per k-half: lo*hi, hi*lo, hi*hi | hi*hi, lo*lo (plain bf16: channels 0-31, 32-63)
        constexpr int NM = NP * TM * TN, NR = 2 * (TM + TN);       // MFMAs and b128 fragment reads per k-half
        struct Frag { f32x4 ah[TM], al[TM], bh[TN], bl[TN]; };
        Frag fr[2];
        // read r of k-half t of the stage whose A / B rows start at Ab / Bb
        auto read_one = [&](Frag& f, int r, unsigned Ab, unsigned Bb, int t) {
            const unsigned ch = (unsigned)(((2 * t + lh) ^ sw) * 16), cl = (unsigned)(((4 + 2 * t + lh) ^ sw) * 16);
            if (r < TM) f.al[r] = lds_read_b128<0>(Ab + cl + r * 32 * BK * 4);
            else if (r < TM + TN) f.bh[r - TM] = lds_read_b128<0>(Bb + ch + (r - TM) * 32 * BK * 4);
            else if (r < 2 * TM + TN) f.ah[r - TM - TN] = lds_read_b128<0>(Ab + ch + (r - TM - TN) * 32 * BK * 4);
            else f.bl[r - 2 * TM - TN] = lds_read_b128<0>(Bb + cl + (r - 2 * TM - TN) * 32 * BK * 4);
        };
        // the NM MFMAs of a k-half in accumulator-rotating order (no two consecutive ones on one accumulator), one fragment read of the
        // next k-half behind each of the first NR; sched_barrier pins the interleave (hipcc would group the reads otherwise)
        auto half = [&](const Frag& cur, Frag& nxt, auto prefetch_tag, unsigned Ab, unsigned Bb, int t_next) {
            constexpr bool prefetch = decltype(prefetch_tag)::value;      // compile time: a run-time test would put a branch behind every MFMA
#pragma unroll
            for (int m = 0; m < (NM > NR ? NM : NR); ++m) {
                if (m < NM) {
                    const int p_ = m / (TM * TN), i = (m % (TM * TN)) / TN, j = m % TN;
                    const bool a_lo = PREC == 1 ? p_ == 0 : p_ == 1, b_lo = PREC == 1 ? p_ == 1 : p_ == 1;
                    const bf16x8 xa = __builtin_bit_cast(bf16x8, a_lo ? cur.al[i] : cur.ah[i]);
                    const bf16x8 xb = __builtin_bit_cast(bf16x8, b_lo ? cur.bl[j] : cur.bh[j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[i][j], 0, 0, 0);
                }
                if (m < NR && prefetch) read_one(nxt, m, Ab, Bb, t_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_s_barrier();                                     // #0: stages 0 and 1 are in LDS
        {
            const unsigned Ab = a_base, Bb = b_base;
#pragma unroll
            for (int r = 0; r < NR; ++r) read_one(fr[0], r, Ab, Bb, 0);  // the only exposed fragment reads of the tile
        }
        int nslot = 1;
        auto stage = [&](int s, auto more_tag) {
            if (s > 0) __builtin_amdgcn_s_barrier();                      // #s: stage s + 1 is in LDS; everyone has finished reading stage s - 1
            const unsigned Ab = a_base + (unsigned)(slot * STAGE * 4), Bb = b_base + (unsigned)(slot * STAGE * 4);
            const unsigned An = a_base + (unsigned)(nslot * STAGE * 4), Bn = b_base + (unsigned)(nslot * STAGE * 4);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            half(fr[0], fr[1], std::true_type{}, Ab, Bb, 1);              // k-half 0 of stage s | reads of its k-half 1
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            half(fr[1], fr[0], more_tag, An, Bn, 0);                      // k-half 1 of stage s | reads of k-half 0 of stage s + 1
            slot = nslot;
            if (++nslot == NS) nslot = 0;
        };
        for (int s = 0; s + 1 < nst; ++s) stage(s, std::true_type{});
        if (nst > 0) stage(nst - 1, std::false_type{});                   // the last stage has nothing to prefetch
    } else {
        for (int s = 0; s < n_main; ++s) iteration(s, std::true_type{});
        for (int s = n_main; s < nst; ++s) iteration(s, std::false_type{});
    }

    VIDC_STAMP(3);      // main loop done
    if (WKW > 1) {
        __syncthreads();
        float* red = smem;
        if (kq > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((kq - 1) * WPK + wq) * TM + i) * TN + j) * 1024 + r * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (kq > 0) return;
#pragma unroll
        for (int q = 1; q < WKW; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += red[((((q - 1) * WPK + wq) * TM + i) * TN + j) * 1024 + r * 64 + lane];
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -----------
    // All flag tests are wave-uniform and hoisted; the arithmetic is branch-free (ReLU off = max with -inf).
    const float lo1 = (a.flags & VIDC_RELU1) ? 0.f : -INFINITY;
    const float lo2 = (a.flags & VIDC_RELU2) ? 0.f : -INFINITY;
    const float lo3 = (a.flags & VIDC_RELU3) ? 0.f : -INFINITY;
    const bool aff2 = a.flags & VIDC_AFFINE2, has_res = a.flags & VIDC_RESIDUAL, accum = a.flags & VIDC_ACCUM;
    const bool st_f32 = !(a.flags & VIDC_NO_F32_OUT), st_split = a.flags & VIDC_SPLIT_OUT;
    unsigned short* ysp = st_split ? a.y_split + (size_t)g * a.y_gs * 2 : nullptr;
    double* stats_out = (PREC == 2 && (a.flags & VIDC_STATS_OUT)) ? reinterpret_cast<double*>(a.y_split) : nullptr;
#ifndef VIDC_CONV_TIMING
    // ---- split-K without a second launch: every k-slice workgroup stores its fp32 partial tile, takes a ticket on the tile's
    //      counter (head of the workspace; device-scope atomic), and the LAST one to arrive sums the splitk partials in slice
    //      order 0..splitk-1 (its own included, re-read from the workspace) and runs the normal epilogue below.  The sum does not
    //      depend on the arrival order, so the result is bit-reproducible; the counter is left at zero for the next launch.
    if (a.splitk > 1) {
        float* part = a.ws + VIDC_SPLITK_COUNTERS;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + wn * 32 * TN + j * 32;
                if (nb >= a.Cout) continue;
                const int mrow = m0 + wm * 32 * TM + i * 32 + 4 * lh;
                float* wsp = part + ((size_t)(kz * a.groups + g) * a.M) * a.Cout + nb + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) __hip_atomic_store(wsp + (size_t)m * a.Cout, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        // The partials travel as relaxed agent-scope atomics (sc1 write-through stores here, sc1 loads below), so they are coherent
        // across the XCDs' L2s without a release/acquire fence pair: an agent-scope fence writes back / invalidates the WHOLE L2 of
        // the XCD (buffer_wbl2 / buffer_inv), which cost ~50 us per launch with HBM-cold weight tiles in flight.  Ordering is by
        // completion instead: every store has been acknowledged (vmcnt 0) before the barrier that precedes the ticket.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                    // (only the WPK epilogue waves are still alive; the ring is free from here on)
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            unsigned* cnt = reinterpret_cast<unsigned*>(a.ws) + ((size_t)g * a.tiles_n + tile_n) * a.tiles_m + tile_m;
            const unsigned ticket = atomicAdd(cnt, 1u);
            const bool last = ticket == (unsigned)(a.splitk - 1);
            if (last) atomicExch(cnt, 0u);
            *flag = last ? 1 : 0;
        }
        __syncthreads();
        if (*reinterpret_cast<volatile int*>(flag) == 0) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + wn * 32 * TN + j * 32;
                if (nb >= a.Cout) continue;
                const int mrow = m0 + wm * 32 * TM + i * 32 + 4 * lh;
                const int n = nb + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                for (int z = 0; z < a.splitk; ++z) {
                    const float* wsp = part + ((size_t)(z * a.groups + g) * a.M) * a.Cout + n;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = min(mrow + (r & 3) + 8 * (r >> 2), a.M - 1);
                        acc[i][j][r] += __hip_atomic_load(wsp + (size_t)m * a.Cout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (has_res) {
                    const int nc = min(n, a.Cout - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = min(mrow + (r & 3) + 8 * (r >> 2), a.M - 1);
                        e_res[i][j][r] = res[(size_t)m * a.ldr + nc];
                    }
                }
            }
    }
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wn * 32 * TN + j * 32;             // first channel of this 32-wide tile (wave-uniform)
            if (nb >= a.Cout) continue;
            const int n = nb + li;
            const int mb = m0 + wm * 32 * TM + i * 32;             // first row of this tile (wave-uniform)
            const int mrow = mb + 4 * lh;
            const bool full = mb + 32 <= a.M;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[i][j][r] * e_s1[j] + e_b1[j], lo1);
            if (aff2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r] * e_s2[j] + e_b2[j], lo2);
            }
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r] + e_res[i][j][r], lo3);
            }
            const unsigned o0 = (unsigned)mrow * (unsigned)a.ldy + (unsigned)n;     // element offset of (mrow, n): M*ldy < 2^31
            // `full_tag` = true_type: all 32 rows of the tile exist (every tile but the last m-tile): no per-row predicate
            auto finish = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                if constexpr (PREC == 2) {
                    // VIDC_STATS_OUT (training, plain-bf16 mode): per-channel sum and sum of squares of this 32-row block of the OUTPUT
                    // (the fp32 values stored below), as the fp64 partials the train-mode BatchNorm behind this conv reduces
                    // (csrc/train.hip chan_final_kernel): saves that BatchNorm's partial-sum launch and its pass over the tensor.
                    // Lane order is fixed (16 rows of this lane in r order, then the lane holding the other 4-row phases), block order
                    // is the chunk index m / 32: bit-reproducible.
                    if (stats_out) {
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dm = (r & 3) + 8 * (r >> 2);
                            if (FULL || mrow + dm < a.M) { const double t = (double)v[r]; s0 += t; s1 += t * t; }
                        }
                        const double o0s = __shfl_xor(s0, 32), o1s = __shfl_xor(s1, 32);
                        if (lh == 0 && n < a.Cout && mb < a.M) {
                            // rows of groups * Cout doubles: a grouped launch writes the partials of ONE BatchNorm over all groups' channels
                            const size_t ctot = (size_t)a.groups * (size_t)a.Cout;
                            double* sp = stats_out + ((size_t)(mb >> 5) * 2) * ctot + (size_t)g * a.Cout + n;
                            sp[0] = s0 + o0s;
                            sp[ctot] = s1 + o1s;
                        }
                    }
                }
                if (accum) {
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        old[r] = (FULL || mrow + dm < a.M) ? yg[o0 + (unsigned)(dm * a.ldy)] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += old[r];
                }
                if (st_f32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        if (FULL || mrow + dm < a.M) yg[o0 + (unsigned)(dm * a.ldy)] = v[r];
                    }
                }
                if (st_split) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        if (FULL || mrow + dm < a.M) store_split(ysp, (size_t)(mrow + dm), a.ldy, n, v[r]);
                    }
                }
            };
            if (full) finish(std::true_type{}); else finish(std::false_type{});
        }
    VIDC_STAMP(4);      // epilogue stores issued
#ifdef VIDC_CONV_TIMING
    if (threadIdx.x == 0) dbg[7] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int BM, int BN, int WMW, int WNW, int WKW, int NS, int PREC, int SPEC>
__global__ void __launch_bounds__(64 * WMW * WNW * WKW * (SPEC ? 2 : 1))
// (Round 4 tried a register budget of 168 for the 2-deep-ring small tiles -- amdgpu_waves_per_eu(3): 172 -> 132 VGPRs, no spills, three
//  workgroups per CU instead of two, so that the 640 workgroups of a layer-3 launch at M = 640 need no second round.  In the frame:
//  per-op times unchanged within 2 % (layer-3 trio 1687 -> 1715 us per tick in fp32), fp32 stream 362 vs 364 frames/s: the work per CU
//  is the same 2.5 workgroups either way.  Not kept; tools/experiments.sh r4_waves3.)
__attribute__((amdgpu_waves_per_eu((BM == 128 && BN == 128 && NS == 2 && !SPEC) ? 2 : 1)))
conv_igemm_f32(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware workgroup mapping (guide T1).  Workgroup w runs on XCD w % 8 (observed, never relied on for
    // correctness: the map below is a bijection).  Each XCD gets a CONTIGUOUS range of the work list ordered
    // (group, n-tile, k-split, m-tile), so one (HBM-cold) weight tile is fetched by one XCD's L2 instead of by all
    // eight; the small, just-produced activation tiles are what gets shared across XCDs.
    int g, tile_n, tile_m, kz;
    {
        const int W = gridDim.x, w = blockIdx.x;
        const int xcd = w & 7, j = w >> 3, q = W >> 3, r = W & 7;
        const unsigned v0 = (unsigned)(xcd * q + min(xcd, r) + j);      // XCDs 0..r-1 own q+1 items, the rest q
        const unsigned v1 = fast_div(v0, a.dv_tiles_m);
        const unsigned v2 = fast_div(v1, a.dv_splitk);
        tile_m = (int)(v0 - v1 * a.tiles_m);
        kz = (int)(v1 - v2 * a.splitk);
        g = (int)fast_div(v2, a.dv_tiles_n);
        tile_n = (int)(v2 - g * a.tiles_n);
    }
    conv_tile<BM, BN, WMW, WNW, WKW, NS, PREC, SPEC>(a, g, tile_m, tile_n, kz, smem);
}

__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                          int KH, int KW) {
    // wp[o][c/32][kh][kw][c%32] = w[o][c][kh][kw]   (K order of the conv kernel: channel unit major, tap minor)
    const long long total = (long long)Cout * Cin * KH * KW;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int K = Cin * KH * KW;
    const int o = (int)(idx / K), k = (int)(idx - (long long)o * K);
    const int ci = k & 31, u = k >> 5, taps = KH * KW;
    const int cu = u / taps, tap = u - cu * taps, kh = tap / KW, kw = tap - kh * KW, c = cu * 32 + ci;
    wp[idx] = w[(((long long)o * Cin + c) * KH + kh) * KW + kw];
}

// ---- bf16x3 operand preparation kernels ---------------------------------------------------------------------------------
// fp32 NHWC rows [rows][ldx] (C channels used) -> [rows][C/32][ hi: 32 x bf16 | lo: 32 x bf16 ]; one thread per 8 channels.
__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long rows,
                                                         int C, int ldx) {
    const int c8 = C / 8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * c8) return;
    const long long r = idx / c8;
    const int c = (int)(idx - r * c8) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c), v1 = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_bf16(v[i], hi[i], lo[i]);
    unsigned short* base = y + (r * (C / 32) + c / 32) * 64 + (c & 31);
    uint4 H, Lo;
    H.x = hi[0] | ((unsigned)hi[1] << 16); H.y = hi[2] | ((unsigned)hi[3] << 16); H.z = hi[4] | ((unsigned)hi[5] << 16); H.w = hi[6] | ((unsigned)hi[7] << 16);
    Lo.x = lo[0] | ((unsigned)lo[1] << 16); Lo.y = lo[2] | ((unsigned)lo[3] << 16); Lo.z = lo[4] | ((unsigned)lo[5] << 16); Lo.w = lo[6] | ((unsigned)lo[7] << 16);
    *reinterpret_cast<uint4*>(base) = H;
    *reinterpret_cast<uint4*>(base + 32) = Lo;
}

// OIHW fp32 -> packed [Cout][Cin/32][KH][KW][hi 32 | lo 32]
__global__ void __launch_bounds__(256) pack_weight_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout,
                                                                 int Cin, int KH, int KW) {
    const long long total = (long long)Cout * Cin * KH * KW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int K = Cin * KH * KW;
    const int o = (int)(idx / K), k = (int)(idx - (long long)o * K);
    const int ci = k & 31, u = k >> 5, taps = KH * KW;
    const int cu = u / taps, tap = u - cu * taps, kh = tap / KW, kw = tap - kh * KW, c = cu * 32 + ci;
    unsigned short hi, lo;
    split_bf16(w[(((long long)o * Cin + c) * KH + kh) * KW + kw], hi, lo);
    unsigned short* base = wp + ((long long)o * (K / 32) + k / 32) * 64 + (k & 31);
    base[0] = hi;
    base[32] = lo;
}

struct TileInfo { int bm, bn, wmw, wnw, wkw, ns; };   // tiles >= kFirstLoaderTile run with loader waves
constexpr TileInfo kTiles[VIDC_TILE_COUNT] = {
    {0, 0, 0, 0, 0, 0},
    {128, 128, 2, 2, 1, 2},   // VIDC_TILE_128x128
    {128, 64, 2, 2, 1, 3},    // VIDC_TILE_128x64
    {64, 128, 2, 2, 1, 3},    // VIDC_TILE_64x128
    {64, 64, 2, 2, 1, 4},     // VIDC_TILE_64x64
    {64, 64, 2, 2, 2, 3},     // VIDC_TILE_64x64_K2   (8 waves)
    {32, 64, 1, 2, 2, 3},     // VIDC_TILE_32x64_K2
    {32, 32, 1, 1, 4, 3},     // VIDC_TILE_32x32_K4
    {32, 128, 1, 4, 1, 4},    // VIDC_TILE_32x128
    {32, 32, 1, 1, 8, 2},     // VIDC_TILE_32x32_K8   (8 waves)
    {32, 64, 1, 2, 2, 5},     // VIDC_TILE_32x64_K2_D5  deeper rings for the DMA-latency-bound small layers
    {32, 32, 1, 1, 4, 4},     // VIDC_TILE_32x32_K4_D4
    {32, 128, 1, 4, 1, 6},    // VIDC_TILE_32x128_D6
    {64, 64, 2, 2, 2, 4},     // VIDC_TILE_64x64_K2_D4
    // ---- with loader waves (SPEC = 1): same tiles, DMA issued by NW extra waves ----
    {32, 64, 1, 2, 2, 3},     // VIDC_TILE_32x64_K2_L
    {32, 64, 1, 2, 2, 5},     // VIDC_TILE_32x64_K2_D5_L
    {32, 32, 1, 1, 4, 4},     // VIDC_TILE_32x32_K4_D4_L
    {64, 64, 2, 2, 1, 4},     // VIDC_TILE_64x64_L
    {64, 64, 2, 2, 2, 4},     // VIDC_TILE_64x64_K2_D4_L   (16 waves)
    {64, 128, 2, 2, 1, 3},    // VIDC_TILE_64x128_L
    {128, 64, 2, 2, 1, 3},    // VIDC_TILE_128x64_L
    // ---- more, narrower weight tiles in flight for the M = 320 layers (HBM-cold weights stream at ~25 B/clk per distinct tile) ----
    {64, 32, 2, 1, 2, 3},     // VIDC_TILE_64x32_K2
    {64, 32, 2, 1, 2, 5},     // VIDC_TILE_64x32_K2_D5
    {64, 32, 2, 1, 2, 5},     // VIDC_TILE_64x32_K2_D5_L
    // ---- 64x64 wave tiles: 0.67 KB of LDS fragment reads per MFMA instead of 1 KB ----
    {128, 128, 2, 2, 1, 3},   // VIDC_TILE_128x128_D3
    {128, 128, 2, 2, 1, 3},   // VIDC_TILE_128x128_D3_L
    {256, 128, 4, 2, 1, 3},   // VIDC_TILE_256x128   (8 waves, 144 KB of LDS)
    {128, 256, 2, 4, 1, 3},   // VIDC_TILE_128x256
    // ---- 2-deep rings: 32-48 KB of LDS, so that >= 3 workgroups (of one or of several streams' launches) share a CU ----
    {32, 64, 1, 2, 2, 2},     // VIDC_TILE_32x64_K2_D2   48 KB
    {64, 64, 2, 2, 1, 2},     // VIDC_TILE_64x64_D2      32 KB
    {32, 32, 1, 1, 4, 2},     // VIDC_TILE_32x32_K4_D2   64 KB
    {64, 128, 2, 2, 1, 2},    // VIDC_TILE_64x128_D2     48 KB
    {64, 32, 2, 1, 2, 2},     // VIDC_TILE_64x32_K2_D2   48 KB
    // ---- loader waves + pipelined fragment reads (SPEC = 2): every ds_read in the shadow of an MFMA, across the stage barrier ----
    {128, 128, 2, 2, 1, 4},   // VIDC_TILE_128x128_D4_P  128 KB
    {128, 128, 2, 2, 1, 3},   // VIDC_TILE_128x128_D3_P   96 KB
    {64, 64, 2, 2, 1, 4},     // VIDC_TILE_64x64_D4_P     64 KB (32x32 wave tiles)
    {128, 64, 2, 2, 1, 4},    // VIDC_TILE_128x64_D4_P    96 KB (64x32 wave tiles)
    {64, 64, 2, 2, 2, 4},     // VIDC_TILE_64x64_K2_D4_P 128 KB (16 waves)
    {64, 32, 2, 1, 2, 5},     // VIDC_TILE_64x32_K2_D5_P 120 KB
    {32, 64, 1, 2, 2, 5},     // VIDC_TILE_32x64_K2_D5_P 120 KB
    // ---- csrc/wgemm.hip: one workgroup streams a chunk of groups of a few-row grouped GEMM through one continuous ring ----
    {96, 32, 3, 1, 2, 2},     // VIDC_TILE_G96x32_STREAM  77 KB
    {96, 64, 3, 2, 2, 3},     // VIDC_TILE_G96x64_STREAM3 145 KB
    {16, 32, 1, 2, 1, 2},     // VIDC_TILE_WINO4_FUSED    144 KB (16 tiles x 32 channels x all 36 positions: csrc/wfused.hip)
};
constexpr int kFirstLoaderTile = VIDC_TILE_32x64_K2_L;

template <int BM, int BN, int WMW, int WNW, int WKW, int NS, int PREC, int SPEC>
int launch_tile_p(const ConvArgs& a, hipStream_t st) {
    constexpr int NT = 64 * WMW * WNW * WKW * (SPEC ? 2 : 1);
    constexpr size_t lds = (size_t)NS * (BM + BN) * BK * WKW * sizeof(float);
    static bool attr_set[64] = {};   // per device (the attribute is per device function); benign race: idempotent
    int dev = 0;
    VIDC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WMW, WNW, WKW, NS, PREC, SPEC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n * a.splitk * a.groups, 1, 1);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WMW, WNW, WKW, NS, PREC, SPEC>), grid, dim3(NT), lds, st, a);
    VIDC_CHECK_LAUNCH("conv_igemm_f32");
    return VIDC_OK;
}

template <int BM, int BN, int WMW, int WNW, int WKW, int NS, int SPEC = 0>
int launch_tile(const ConvArgs& a, hipStream_t st, int precision) {
    return precision == VIDC_PREC_BF16X3 ? launch_tile_p<BM, BN, WMW, WNW, WKW, NS, 1, SPEC>(a, st)
           : precision == VIDC_PREC_BF16 ? launch_tile_p<BM, BN, WMW, WNW, WKW, NS, 2, SPEC>(a, st)
                                         : launch_tile_p<BM, BN, WMW, WNW, WKW, NS, 0, SPEC>(a, st);
}

int validate(const vidc_conv_desc* d) {
    VIDC_REQUIRE(d, VIDC_ERR_NULL, "conv: null descriptor");
    VIDC_REQUIRE(d->x && d->w && d->y && d->scale1 && d->shift1, VIDC_ERR_NULL, "conv: null tensor pointer");
    VIDC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, VIDC_ERR_SHAPE, "conv: bad spatial dims");
    VIDC_REQUIRE(d->Cin > 0 && d->Cin % BK == 0, VIDC_ERR_SHAPE, "conv: Cin=%d must be a multiple of %d", d->Cin, BK);
    // whole 32-channel MFMA tiles only: the epilogue / split-K stores guard per tile, not per lane (include/vidc.h documents Cout % 32 == 0)
    VIDC_REQUIRE(d->Cout > 0 && d->Cout % 32 == 0, VIDC_ERR_SHAPE, "conv: Cout=%d must be a multiple of 32", d->Cout);
    VIDC_REQUIRE(d->ldx >= d->Cin && d->ldy >= d->Cout && d->ldx % 4 == 0, VIDC_ERR_SHAPE, "conv: bad channel strides");
    VIDC_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH <= 3 && d->KW <= 3 && d->stride >= 1 && d->pad >= 0, VIDC_ERR_SHAPE,
                 "conv: bad kernel geometry (kernels up to 3x3)");
    {
        const int dil = d->dilation > 1 ? d->dilation : 1;
        VIDC_REQUIRE(d->dilation >= 0 && d->Ho == (d->H + 2 * d->pad - dil * (d->KH - 1) - 1) / d->stride + 1 &&
                         d->Wo == (d->W + 2 * d->pad - dil * (d->KW - 1) - 1) / d->stride + 1,
                     VIDC_ERR_SHAPE, "conv: Ho/Wo inconsistent with H/W, kernel, stride, pad, dilation");
    }
    VIDC_REQUIRE(d->groups >= 1 && d->splitk >= 1, VIDC_ERR_SHAPE, "conv: groups/splitk must be >= 1");
    VIDC_REQUIRE(!(d->flags & VIDC_AFFINE2) || (d->scale2 && d->shift2), VIDC_ERR_NULL, "conv: AFFINE2 without scale2/shift2");
    VIDC_REQUIRE(!(d->flags & VIDC_RESIDUAL) || (d->residual && d->ldr >= d->Cout), VIDC_ERR_NULL, "conv: RESIDUAL without tensor");
    VIDC_REQUIRE(d->tile >= 0 && d->tile < VIDC_TILE_COUNT, VIDC_ERR_SHAPE, "conv: unknown tile id %d", d->tile);
    VIDC_REQUIRE(d->precision == VIDC_PREC_FP32 || d->precision == VIDC_PREC_BF16X3 || d->precision == VIDC_PREC_BF16, VIDC_ERR_SHAPE,
                 "conv: unknown precision %d", d->precision);
    VIDC_REQUIRE(d->precision != VIDC_PREC_BF16 || !(d->flags & VIDC_SPLIT_OUT), VIDC_ERR_SHAPE, "conv: SPLIT_OUT writes the bf16x3 format, not plain bf16");
    VIDC_REQUIRE(d->splitk == 1 || d->workspace || d->tile == VIDC_TILE_G96x32_STREAM || d->tile == VIDC_TILE_G96x64_STREAM3, VIDC_ERR_NULL, "conv: split-K needs a workspace");
    VIDC_REQUIRE(!(d->flags & VIDC_SPLIT_OUT) || (d->y_split && d->Cout % 32 == 0 && d->ldy % 32 == 0), VIDC_ERR_NULL,
                 "conv: SPLIT_OUT needs y_split and Cout, ldy multiples of 32");
    VIDC_REQUIRE(!(d->flags & VIDC_NO_F32_OUT) || (d->flags & VIDC_SPLIT_OUT), VIDC_ERR_SHAPE, "conv: NO_F32_OUT without SPLIT_OUT writes nothing");
    VIDC_REQUIRE(!(d->flags & VIDC_STATS_OUT) || (d->precision == VIDC_PREC_BF16 && d->y_split &&
                                                  !(d->flags & (VIDC_AFFINE2 | VIDC_RESIDUAL | VIDC_ACCUM | VIDC_SPLIT_OUT | VIDC_NO_F32_OUT))),
                 VIDC_ERR_SHAPE, "conv: STATS_OUT needs VIDC_PREC_BF16, y_split = the partials buffer and a plain epilogue");
    VIDC_REQUIRE(!(d->flags & VIDC_X_PLANAR_GROUPS) || (long long)d->groups * d->x_gs * 4 < (1ll << 31), VIDC_ERR_SHAPE,
                 "conv: X_PLANAR_GROUPS: the planes of all groups must stay below 2 GiB");
    VIDC_REQUIRE((long long)d->B * d->Ho * d->Wo < (1ll << 31), VIDC_ERR_SHAPE, "conv: M overflows int32");
    VIDC_REQUIRE((long long)d->Cout * d->KH * d->KW * d->Cin * 4 < (1ll << 31), VIDC_ERR_SHAPE,
                 "conv: one group's weights must stay below 2 GiB (32-bit buffer offsets)");
    VIDC_REQUIRE((long long)d->B * d->Ho * d->Wo * d->ldy < (1ll << 31) && (long long)d->B * d->H * d->W * d->ldx < (1ll << 29),
                 VIDC_ERR_SHAPE, "conv: tensor too large for 32-bit offsets (M*ldy < 2^31 elements, input < 2 GiB)");
    return VIDC_OK;
}

// Kernel arguments of one conv for tile `dd.tile` / `dd.splitk` (both already chosen).
int make_args(const vidc_conv_desc& dd, ConvArgs& a) {
    a.x = dd.x; a.w = dd.w; a.y = dd.y; a.scale1 = dd.scale1; a.shift1 = dd.shift1; a.scale2 = dd.scale2; a.shift2 = dd.shift2;
    a.residual = dd.residual; a.ws = dd.workspace; a.y_split = reinterpret_cast<unsigned short*>(dd.y_split);
    a.B = dd.B; a.H = dd.H; a.W = dd.W; a.Cin = dd.Cin; a.ldx = dd.ldx; a.Ho = dd.Ho; a.Wo = dd.Wo; a.Cout = dd.Cout;
    a.ldy = dd.ldy; a.ldr = dd.ldr; a.KH = dd.KH; a.KW = dd.KW; a.stride = dd.stride; a.pad = dd.pad; a.flags = dd.flags;
    a.groups = dd.groups; a.dil = dd.dilation > 1 ? dd.dilation : 1; a.x_gs = dd.x_gs; a.w_gs = dd.w_gs; a.y_gs = dd.y_gs; a.r_gs = dd.r_gs; a.p_gs = dd.p_gs;
    a.M = dd.B * dd.Ho * dd.Wo; a.K = dd.KH * dd.KW * dd.Cin; a.ksteps = a.K / BK;
    a.splitk = dd.splitk;
    const TileInfo ti = kTiles[dd.tile];
    a.tiles_m = (a.M + ti.bm - 1) / ti.bm;
    a.tiles_n = (a.Cout + ti.bn - 1) / ti.bn;
    {
        const int stages = (a.ksteps + ti.wkw - 1) / ti.wkw;
        if (a.splitk > stages) a.splitk = stages;
        if ((long long)a.tiles_m * a.tiles_n * a.groups > VIDC_SPLITK_COUNTERS) a.splitk = 1;    // one ticket counter per output tile
    }
    fast_div_init((unsigned)a.tiles_m, a.dv_tiles_m);
    fast_div_init((unsigned)a.splitk, a.dv_splitk);
    fast_div_init((unsigned)a.tiles_n, a.dv_tiles_n);
    fast_div_init((unsigned)(a.KH * a.KW), a.dv_ntaps);
    fast_div_init((unsigned)a.KW, a.dv_kw);
    VIDC_REQUIRE((long long)a.tiles_m * a.tiles_n * a.splitk * a.groups < (1ll << 31), VIDC_ERR_SHAPE, "conv: grid too large");
    return VIDC_OK;
}

}  // namespace

extern "C" int vidc_conv2d_plan(vidc_conv_desc* d) {
    VIDC_REQUIRE(d, VIDC_ERR_NULL, "conv plan: null descriptor");
    // Cost model (the measured table in conv_tuning.json overrides it): every SIMD issues one 32x32x2 MFMA per 64 clk.
    // A workgroup of NW waves puts NW/4 waves on each SIMD of its CU; with 256 CUs a launch takes about
    // ceil(workgroups / 256) * (MFMA clocks of one workgroup per SIMD) + a fixed fill/drain cost, and global split-K
    // adds a pass over the fp32 partials plus a second launch.
    const long long M = (long long)d->B * d->Ho * d->Wo;
    const int units = d->KH * d->KW * d->Cin / BK;
    const int n_cu = 256;
    double best = 1e30;
    int best_tile = VIDC_TILE_64x64, best_sk = 1;
    for (int t = 1; t < kFirstLoaderTile; ++t) {      // loader-wave variants are chosen by the measured table only
        const TileInfo ti = kTiles[t];
        if (ti.bn > d->Cout && ti.bn > 64) continue;
        const long long tm = (M + ti.bm - 1) / ti.bm, tn = (d->Cout + ti.bn - 1) / ti.bn;
        const int nw = ti.wmw * ti.wnw * ti.wkw;
        const double mfma_per_wave_stage = (double)(ti.bm / (32 * ti.wmw)) * (ti.bn / (32 * ti.wnw)) * 16.0;
        const int stages = (units + ti.wkw - 1) / ti.wkw;
        for (int sk = 1; sk <= 16; sk *= 2) {
            if (sk > 1 && stages / sk < 4) break;
            const long long wgs = tm * tn * sk * d->groups;
            const double rounds = ceil((double)wgs / n_cu);
            double cyc = rounds * ((double)((stages + sk - 1) / sk) * mfma_per_wave_stage * 64.0 * (nw / 4.0) + 2500.0) + 4000.0;
            if (ti.wkw > 1) cyc += 1500.0;
            if (sk > 1) cyc += 9000.0 + (double)M * d->Cout * d->groups * (sk + 1) * 4.0 / 2500.0;   // ~6 TB/s @2.4 GHz
            if (cyc < best) { best = cyc; best_tile = t; best_sk = sk; }
        }
    }
    d->tile = best_tile;
    d->splitk = best_sk;
    return VIDC_OK;
}

extern "C" size_t vidc_conv2d_workspace_bytes(const vidc_conv_desc* d) {
    if (!d || d->splitk <= 1 || d->tile == VIDC_TILE_G96x32_STREAM || d->tile == VIDC_TILE_G96x64_STREAM3) return 0;      // (streamed tile: splitk = its number of group chunks, no partials)
    return ((size_t)VIDC_SPLITK_COUNTERS + (size_t)d->splitk * d->groups * d->B * d->Ho * d->Wo * d->Cout) * sizeof(float);
}

extern "C" int vidc_conv2d_bn_act(const vidc_conv_desc* d, vidc_stream_t stream) {
    int rc = validate(d);
    if (rc != VIDC_OK) return rc;
    vidc_conv_desc dd = *d;
    if (dd.tile == VIDC_TILE_AUTO) {
        int sk = dd.splitk;
        vidc_conv2d_plan(&dd);
        dd.splitk = dd.workspace ? sk : 1;      // the caller sized the workspace for ITS splitk: the planner only picks the tile here
    }
    ConvArgs a;
    rc = make_args(dd, a);
    if (rc != VIDC_OK) return rc;
    hipStream_t st = vidc::as_stream(stream);
    switch (dd.tile) {
        case VIDC_TILE_128x128:  rc = launch_tile<128, 128, 2, 2, 1, 2>(a, st, dd.precision); break;
        case VIDC_TILE_128x64:   rc = launch_tile<128, 64, 2, 2, 1, 3>(a, st, dd.precision); break;
        case VIDC_TILE_64x128:   rc = launch_tile<64, 128, 2, 2, 1, 3>(a, st, dd.precision); break;
        case VIDC_TILE_64x64:    rc = launch_tile<64, 64, 2, 2, 1, 4>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_K2: rc = launch_tile<64, 64, 2, 2, 2, 3>(a, st, dd.precision); break;
        case VIDC_TILE_32x64_K2: rc = launch_tile<32, 64, 1, 2, 2, 3>(a, st, dd.precision); break;
        case VIDC_TILE_32x32_K4: rc = launch_tile<32, 32, 1, 1, 4, 3>(a, st, dd.precision); break;
        case VIDC_TILE_32x128:   rc = launch_tile<32, 128, 1, 4, 1, 4>(a, st, dd.precision); break;
        case VIDC_TILE_32x32_K8: rc = launch_tile<32, 32, 1, 1, 8, 2>(a, st, dd.precision); break;
        case VIDC_TILE_32x64_K2_D5: rc = launch_tile<32, 64, 1, 2, 2, 5>(a, st, dd.precision); break;
        case VIDC_TILE_32x32_K4_D4: rc = launch_tile<32, 32, 1, 1, 4, 4>(a, st, dd.precision); break;
        case VIDC_TILE_32x128_D6:   rc = launch_tile<32, 128, 1, 4, 1, 6>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_K2_D4: rc = launch_tile<64, 64, 2, 2, 2, 4>(a, st, dd.precision); break;
        case VIDC_TILE_32x64_K2_L:     rc = launch_tile<32, 64, 1, 2, 2, 3, 1>(a, st, dd.precision); break;
        case VIDC_TILE_32x64_K2_D5_L:  rc = launch_tile<32, 64, 1, 2, 2, 5, 1>(a, st, dd.precision); break;
        case VIDC_TILE_32x32_K4_D4_L:  rc = launch_tile<32, 32, 1, 1, 4, 4, 1>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_L:        rc = launch_tile<64, 64, 2, 2, 1, 4, 1>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_K2_D4_L:  rc = launch_tile<64, 64, 2, 2, 2, 4, 1>(a, st, dd.precision); break;
        case VIDC_TILE_64x128_L:       rc = launch_tile<64, 128, 2, 2, 1, 3, 1>(a, st, dd.precision); break;
        case VIDC_TILE_128x64_L:       rc = launch_tile<128, 64, 2, 2, 1, 3, 1>(a, st, dd.precision); break;
        case VIDC_TILE_64x32_K2:       rc = launch_tile<64, 32, 2, 1, 2, 3>(a, st, dd.precision); break;
        case VIDC_TILE_64x32_K2_D5:    rc = launch_tile<64, 32, 2, 1, 2, 5>(a, st, dd.precision); break;
        case VIDC_TILE_64x32_K2_D5_L:  rc = launch_tile<64, 32, 2, 1, 2, 5, 1>(a, st, dd.precision); break;
        case VIDC_TILE_128x128_D3:     rc = launch_tile<128, 128, 2, 2, 1, 3>(a, st, dd.precision); break;
        case VIDC_TILE_128x128_D3_L:   rc = launch_tile<128, 128, 2, 2, 1, 3, 1>(a, st, dd.precision); break;
        case VIDC_TILE_256x128:        rc = launch_tile<256, 128, 4, 2, 1, 3>(a, st, dd.precision); break;
        case VIDC_TILE_128x256:        rc = launch_tile<128, 256, 2, 4, 1, 3>(a, st, dd.precision); break;
        case VIDC_TILE_32x64_K2_D2:    rc = launch_tile<32, 64, 1, 2, 2, 2>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_D2:       rc = launch_tile<64, 64, 2, 2, 1, 2>(a, st, dd.precision); break;
        case VIDC_TILE_32x32_K4_D2:    rc = launch_tile<32, 32, 1, 1, 4, 2>(a, st, dd.precision); break;
        case VIDC_TILE_64x128_D2:      rc = launch_tile<64, 128, 2, 2, 1, 2>(a, st, dd.precision); break;
        case VIDC_TILE_64x32_K2_D2:    rc = launch_tile<64, 32, 2, 1, 2, 2>(a, st, dd.precision); break;
        case VIDC_TILE_128x128_D4_P:   rc = launch_tile<128, 128, 2, 2, 1, 4, 2>(a, st, dd.precision); break;
        case VIDC_TILE_128x128_D3_P:   rc = launch_tile<128, 128, 2, 2, 1, 3, 2>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_D4_P:     rc = launch_tile<64, 64, 2, 2, 1, 4, 2>(a, st, dd.precision); break;
        case VIDC_TILE_128x64_D4_P:    rc = launch_tile<128, 64, 2, 2, 1, 4, 2>(a, st, dd.precision); break;
        case VIDC_TILE_64x64_K2_D4_P:  rc = launch_tile<64, 64, 2, 2, 2, 4, 2>(a, st, dd.precision); break;
        case VIDC_TILE_64x32_K2_D5_P:  rc = launch_tile<64, 32, 2, 1, 2, 5, 2>(a, st, dd.precision); break;
        case VIDC_TILE_32x64_K2_D5_P:  rc = launch_tile<32, 64, 1, 2, 2, 5, 2>(a, st, dd.precision); break;
        case VIDC_TILE_G96x32_STREAM:
        case VIDC_TILE_G96x64_STREAM3: rc = vidc::launch_wgemm_stream(dd, st); break;
        case VIDC_TILE_WINO4_FUSED:    rc = vidc::launch_wino4_fused(dd, st); break;
        default: VIDC_REQUIRE(false, VIDC_ERR_SHAPE, "conv: bad tile");
    }
    return rc;
}

extern "C" int vidc_split_bf16x3(const float* x, void* y, long long rows, int C, int ldx, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y, VIDC_ERR_NULL, "vidc_split_bf16x3: null pointer");
    VIDC_REQUIRE(rows > 0 && C > 0 && C % 32 == 0 && ldx >= C && ldx % 4 == 0, VIDC_ERR_SHAPE, "vidc_split_bf16x3: C must be a multiple of 32");
    const long long n = rows * (C / 8);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), x,
                       reinterpret_cast<unsigned short*>(y), rows, C, ldx);
    VIDC_CHECK_LAUNCH("split_rows_kernel");
    return VIDC_OK;
}

extern "C" int vidc_pack_conv_weight_bf16x3(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, vidc_stream_t stream) {
    VIDC_REQUIRE(w_oihw && w_packed, VIDC_ERR_NULL, "vidc_pack_conv_weight_bf16x3: null pointer");
    VIDC_REQUIRE(Cout > 0 && Cin > 0 && Cin % 32 == 0 && KH > 0 && KW > 0, VIDC_ERR_SHAPE, "vidc_pack_conv_weight_bf16x3: Cin must be a multiple of 32");
    const long long total = (long long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), w_oihw,
                       reinterpret_cast<unsigned short*>(w_packed), Cout, Cin, KH, KW);
    VIDC_CHECK_LAUNCH("pack_weight_bf16x3_kernel");
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
