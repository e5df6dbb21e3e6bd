// Winograd F(4x4, 3x3) in ONE launch for the small maps: input transform, the 36 transform-domain products and the output transform with the conv's
// epilogue, without the V and M tensors (VERDICT r5 items 2 and 3 iii).  BUILT, PARITY-GREEN AND NOT FASTER: the engine does not use it (see the end).
//
// On a 16x20 map (ResNet-101 layer 3 at the program batch bench.py times: 4 frames x 4 x 5 tiles = 80 tiles per pyramid, Cin = Cout = 256,
// networks/surface_normal.py:27-50) the three launches of csrc/winograd.hip + conv_mfma.hip take 7 + 29 + 9.5 us, of which three launch floors and the
// lock-step stage skeleton of a few-row GEMM are more than half: the products are 1.5 GFLOP = 9.6 us of the chip's fp32 MFMA rate.  Here a
// workgroup owns 16 tiles x 32 output channels of one group and ALL 36 positions (12 waves, 144 KB of LDS, one workgroup per CU):
//
//   waves 8..11 (transform): thread = (tile, channel pair); per chunk of 32 input channels it loads the tile's 6 x 6 patch (36 unconditional 8-byte
//       buffer loads, out-of-map taps encoded as out-of-range offsets), computes V = B^T d B in registers (the formulas of csrc/winograd.hip, no
//       contraction, three transformed rows at a time) and writes [pos][tile][32 ch] into the LDS buffer that the product waves are NOT reading (two
//       buffers of 72 KB, one barrier per chunk); the loads of chunk kc + 2 are issued in the middle of the transform of chunk kc + 1.
//   waves 0..7 (products): wave w < 4 owns positions 5 w .. 5 w + 4, wave w >= 4 positions 20 + 4 (w - 4) ..: waves w and w + 4 share a SIMD = 9
//       positions per SIMD, and one of the two issues MFMAs while the other is held up issuing a load (a VMEM instruction costs its wave ~75 clk
//       whatever its width -- with ONE product wave per SIMD the U loads and the MFMAs simply added up: 56 us).  Per position 2 accumulators of
//       v_mfma_f32_16x16x4_f32 (16 tiles x 16 channels each) over all of K.  A = V from LDS (two ds_read_b128 per position and chunk, rows XOR-
//       swizzled: conflict-free without padding); B = U straight from global memory / L2 into registers in FRAGMENT ORDER
//       (vidc_winograd_weight_pack_fused: every load instruction is 1 KiB contiguous; MFMA k-step ks of lane group kq multiplies channel 8 kq + ks of
//       the chunk -- the K order is free as long as A and B agree), one whole chunk ahead, across chunk boundaries and barriers.
//   epilogue: the 36 x 16 x 32 products go through LDS once, thread = (tile, channel) folds them with A^T (.) A, applies the conv's affine / ReLU
//       [/ second affine / ReLU] and stores its 4 x 4 pixels (128 contiguous bytes per pixel and workgroup).
//
// Workgroup L = (tile block, group, channel block) with the (group, channel block) pair fastest: the tile blocks that stream the same slice of U
// share L % 8 = one XCD's L2.  Result bits of a tile depend on nothing but the tile (fixed K order, fixed fold): a restriction to fewer groups
// (engine.Program.group_variant) or another batch leaves an item's bits alone (tests/test_wfused.py).
//
// Measured (tools/wfused_bench.py, weights HBM-cold, profiles/r6_wfused_bench.txt): layer 3 (4 groups) 46.9 us against 7.2 + 29.0 + 9.6 = 45.7 us for the
// three launches; layer 2 48.9 / 40.2; one group alone 39.8 / 28.9.  Attribution builds (make wfused_attrib, tools/wfused_attrib.sh): empty skeleton
// 10.5 us (= the launch floor), + MFMAs 28.9 (18.4 us of products: 160 workgroups use 160 of 256 CUs and v_mfma_f32_16x16x4_f32 reads twice the operand
// bytes per FLOP of the 32x32 form), + U loads 35.6, + transform 46.5 (of which the fold 4).  Without its MFMAs the launch still takes 40.4 us: once the
// transform wave of a SIMD is slower than the SIMD's products (36 VMEM instructions + ~500 VALU per chunk against 2.3 us of MFMA) its patch loads have
// no product phase left to hide under and every chunk pays their latency.  Making the transform cheaper needs registers (a second patch in flight, or
// 16-byte loads: 144 + registers per thread against the 170 of a 12-wave workgroup) or LDS (full: the two V buffers are 144 of 160 KB).  Equal time,
// so the measured table keeps the three-launch form; the kernel stays in the library as tile VIDC_TILE_WINO4_FUSED with its tests.
#include "common.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 16, NBW = 32, KC = 32, NPOS = 36;
constexpr int NPW = 8, NTW = 4, NTHREADS = 64 * (NPW + NTW);      // 8 product waves (5 or 4 positions each: 9 per SIMD), 4 transform waves
constexpr int VBUF = NPOS * TB * KC;                 // floats of one V buffer
constexpr int MLD = 36;                              // row stride of the product rows in the epilogue (floats): 4 kq row groups -> 4 bank groups
constexpr size_t LDS_BYTES = (size_t)2 * VBUF * sizeof(float);
static_assert((size_t)NPOS * TB * MLD * sizeof(float) <= LDS_BYTES, "the epilogue's product rows reuse the V buffers");

struct FArgs {
    const float* x; const float* u; float* y; const float* s1; const float* b1; const float* s2; const float* b2;
    int H, W, Cin, ldx, Cout, ldy, th, tw, T, flags, nbn, gn;      // T = B * th * tw tiles; nbn = Cout / 32; gn = groups * nbn
    long long x_gs, w_gs, y_gs, p_gs;
    unsigned x_bytes;
};

// B^T d for one column / A^T m for one column: Lavin & Gray's F(4x4, 3x3) matrices, the operation order of csrc/winograd.hip Wino<4>
template <typename V> __device__ __forceinline__ void bt6(const V (&d)[6], V (&t)[6]) {
    const V p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1];
    const V r = d[4] - d[2], s = 2.f * (d[3] - d[1]);
    t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
    t[1] = p + q;
    t[2] = p - q;
    t[3] = r + s;
    t[4] = r - s;
    t[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
}
// the same in two halves (rows 0..2 / 3..5 of B^T d share no intermediate): the transform waves keep three transformed rows at a time
template <typename V> __device__ __forceinline__ void bt6_lo(const V (&d)[6], V (&t)[3]) {
    const V p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1];
    t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
    t[1] = p + q;
    t[2] = p - q;
}
template <typename V> __device__ __forceinline__ void bt6_hi(const V (&d)[6], V (&t)[3]) {
    const V r = d[4] - d[2], s = 2.f * (d[3] - d[1]);
    t[0] = r + s;
    t[1] = r - s;
    t[2] = (4.f * d[1] - 5.f * d[3]) + d[5];
}
__device__ __forceinline__ void at6(const float (&m)[6], float (&o)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = (m[0] + s12) + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = (d12 + 8.f * d34) + m[5];
}

template <int N_> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory"); }
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct V2 { float x, y; };
__device__ __forceinline__ V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
__device__ __forceinline__ void lds_write_b64(unsigned addr, V2 v) {
    const f32x2 t = {v.x, v.y};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(t) : "memory");
}

// One product wave: positions pos0 .. pos0 + PW - 1, all of K, 16 tiles x 32 output channels.
template <int PW, int DBG>
__device__ __forceinline__ void product_wave(const FArgs& a, float* smem, unsigned lds0, int lane, int pos0, int g, int nb, int NK) {
    {
        const int n = lane & 15, kq = lane >> 4;
        // U in fragment order (vidc_winograd_weight_pack_fused): [g][pos][nb][kc][nblk][h][lane][4] -- one load instruction = 1 KiB contiguous
        const float* ub = a.u + (size_t)g * a.w_gs + ((size_t)pos0 * a.nbn + nb) * ((size_t)NK * 1024) + lane * 4;
        const size_t pos_stride = (size_t)a.nbn * NK * 1024;
        auto load_b = [&](int p, int kc, f32x4 (&dst)[2][2]) {
            const float* s = ub + (size_t)p * pos_stride + (size_t)kc * 1024;
            dst[0][0] = *reinterpret_cast<const f32x4*>(s);
            dst[0][1] = *reinterpret_cast<const f32x4*>(s + 256);
            dst[1][0] = *reinterpret_cast<const f32x4*>(s + 512);
            dst[1][1] = *reinterpret_cast<const f32x4*>(s + 768);
        };
        f32x4 acc[PW][2];
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bq[PW][2][2];                           // a whole chunk of the wave's U fragments ahead (12 KB per wave, 144 KB per CU in flight)
#pragma unroll
        for (int p = 0; p < PW; ++p) load_b(p, 0, bq[p]);
        // A fragment of (position, tile m = lane & 15): channels 8 kq .. 8 kq + 7 of the row = 16-byte units 2 kq, 2 kq + 1, unit index XOR (m >> 1) & 7
        const int sw = (n >> 1) & 7;
        const unsigned a_off0 = (unsigned)((n * KC + (((kq * 2) ^ sw) * 4)) * 4), a_off1 = (unsigned)((n * KC + (((kq * 2 + 1) ^ sw) * 4)) * 4);
        const unsigned a_wave = lds0 + (unsigned)(pos0 * TB * KC * 4);

        __builtin_amdgcn_s_barrier();                 // chunk 0 of V is in buffer 0
        for (int kc = 0; kc < NK; ++kc) {
            const unsigned vb = a_wave + (unsigned)((kc & 1) * VBUF * 4);
            const int kn = kc + 1 < NK ? kc + 1 : kc;      // (the look-ahead loads of the last chunk re-read it: harmless, no tail case)
            f32x4 fa[2][2];
            fa[0][0] = lds_read_b128(vb + a_off0);
            fa[0][1] = lds_read_b128(vb + a_off1);
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                const int cur = p & 1, nxt = cur ^ 1;
                if (p + 1 < PW) {
                    fa[nxt][0] = lds_read_b128(vb + (unsigned)((p + 1) * TB * KC * 4) + a_off0);
                    fa[nxt][1] = lds_read_b128(vb + (unsigned)((p + 1) * TB * KC * 4) + a_off1);
                    wait_lgkmcnt<2>();
                } else {
                    wait_lgkmcnt<0>();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if constexpr (!(DBG & 1)) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][h][ks], bq[p][j][h][ks], acc[p][j], 0, 0, 0);
                            else if (h == 0 && ks == 0) acc[p][j][0] += fa[cur][0][0] + fa[cur][1][3] + bq[p][j][0][0] + bq[p][j][1][3];
                if constexpr (!(DBG & 2)) load_b(p, kn, bq[p]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();             // every wave has read buffer kc & 1; chunk kc + 1 is in the other one
        }
        // products -> LDS rows [pos][tile][MLD]: lane holds rows 4 kq + i, column n of each 16 x 16 block
        float* mx = smem;
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx[((pos0 + p) * TB + 4 * kq + i) * MLD + j * 16 + n] = acc[p][j][i];
    }
}

// DBG (attribution builds, -DVIDC_WFUSED_ATTRIB + VIDC_WFUSED_DBG): 1 no MFMAs, 2 no U loads in the loop, 4 no transform (barriers only), 8 no fold / stores
template <int DBG>
__global__ void __launch_bounds__(NTHREADS)
wino4_fused_kernel(const FArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int L = blockIdx.x;
    const int tb = L / a.gn, r_ = L - tb * a.gn, g = r_ / a.nbn, nb = r_ - g * a.nbn;
    const int NK = a.Cin / KC;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) float*)smem);

    if (tid < 64 * NPW) {
        // product waves: wave w and wave w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs in turn); 5 + 4 positions = 9 per SIMD
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (w < 4) product_wave<5, DBG>(a, smem, lds0, tid & 63, 5 * w, g, nb, NK);
        else product_wave<4, DBG>(a, smem, lds0, tid & 63, 20 + 4 * (w - 4), g, nb, NK);
    } else {
        // ------------------------------------------------------------------ transform waves: thread = (tile m, channel pair q)
        const int tt = tid - 64 * NPW, m = tt >> 4, q = tt & 15;
        const int t = tb * TB + m;
        const bool tv = t < a.T;
        const int tpf = a.th * a.tw;
        const int b = tv ? t / tpf : 0, rem = tv ? t - b * tpf : 0, ty = rem / a.tw, tx = rem - ty * a.tw;
        const int iy0 = ty * 4 - 1, ix0 = tx * 4 - 1;
        // patch addressing: byte offsets into ONE buffer descriptor over x (below 1 GiB: the launcher checks); a row outside the map (or a tile beyond T)
        // contributes 2^31, a column outside 2^30 -- the sum is beyond the descriptor's range and the load returns zeros: 36 unconditional 8-byte loads
        // per chunk, offset = base + row term + column term, no select, no branch.  (Wide loads on purpose: a VMEM instruction costs the wave ~75 clk
        // whatever its width -- one channel per thread and load made the transform the longest phase of the kernel.)
        unsigned ro[6], co[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int iy = iy0 + i, ix = ix0 + i;
            const bool ry = tv && (unsigned)iy < (unsigned)a.H, cx = (unsigned)ix < (unsigned)a.W;
            ro[i] = ry ? (unsigned)(((b * a.H + iy) * a.W) * a.ldx) * 4u : 0x80000000u;
            co[i] = cx ? (unsigned)(ix * a.ldx) * 4u : 0x40000000u;
        }
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const unsigned xb = (unsigned)(g * a.x_gs + 2 * q) * 4u;
        V2 D[36], tcol[3][6];
        auto load_d = [&](int kc) {
            const int soff = kc * KC * 4;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                unsigned rb = xb + ro[r];
                asm volatile("" : "+v"(rb));          // (keeps the 36 sums out of registers: hipcc would hoist them out of the chunk loop and spill)
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, (int)(rb + co[s]), soff, 0);
                    D[r * 6 + s] = V2{__uint_as_float(v.x), __uint_as_float(v.y)};
                }
            }
        };
        // B^T d (columns), three of its six rows at a time, then (.) B (rows) of those three to LDS; after the second column pass D is dead and takes the
        // NEXT chunk's loads: one patch in registers, the next one in flight for a whole product phase
        auto col_pass = [&](int half) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                V2 col[6], out[3];
#pragma unroll
                for (int r = 0; r < 6; ++r) col[r] = D[r * 6 + s];
                if (half == 0) bt6_lo(col, out);
                else bt6_hi(col, out);
#pragma unroll
                for (int r = 0; r < 3; ++r) tcol[r][s] = out[r];
            }
        };
        // row (pos, m) of buffer `buf`, channels 2 q, 2 q + 1: unit (q >> 1) ^ ((m >> 1) & 7), second half of the unit for odd q
        const unsigned w_off = lds0 + (unsigned)((m * KC + (((q >> 1) ^ ((m >> 1) & 7)) * 4) + (q & 1) * 2) * 4);
        auto row_pass = [&](int buf, int half) {
            const unsigned base = w_off + (unsigned)(buf * VBUF * 4);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                V2 out[6];
                bt6(tcol[r], out);
#pragma unroll
                for (int s = 0; s < 6; ++s) lds_write_b64(base + (unsigned)(((3 * half + r) * 6 + s) * TB * KC * 4), out[s]);
            }
        };
        load_d(0);
        col_pass(0);
        row_pass(0, 0);
        col_pass(1);
        load_d(NK > 1 ? 1 : 0);
        row_pass(0, 1);
        wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int kc = 0; kc < NK; ++kc) {             // iteration kc: chunk kc + 1 -> buffer (kc + 1) & 1 while the product waves read buffer kc & 1
            if (kc + 1 < NK && !(DBG & 4)) {
                col_pass(0);
                row_pass((kc + 1) & 1, 0);
                col_pass(1);
                load_d(kc + 2 < NK ? kc + 2 : NK - 1);
                row_pass((kc + 1) & 1, 1);
            }
            wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------------- fold + epilogue: thread = (tile m, channel c)
    if (tid < TB * NBW) {
        const int m = tid >> 5, c = tid & 31;
        const int t = tb * TB + m;
        if (t >= a.T || (DBG & 8)) return;
        const int tpf = a.th * a.tw;
        const int b = t / tpf, rem = t - b * tpf, ty = rem / a.tw, tx = rem - ty * a.tw;
        const float* mx = smem + m * MLD + c;
        float tc[4][6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float col[6], out[4];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = mx[(r * 6 + s) * TB * MLD];
            at6(col, out);
#pragma unroll
            for (int r = 0; r < 4; ++r) tc[r][s] = out[r];
        }
        const int ch = nb * NBW + c;
        const float s1 = a.s1[(size_t)g * a.p_gs + ch], b1 = a.b1[(size_t)g * a.p_gs + ch];
        const bool aff2 = a.flags & VIDC_AFFINE2;
        const float s2 = aff2 ? a.s2[(size_t)g * a.p_gs + ch] : 0.f, b2 = aff2 ? a.b2[(size_t)g * a.p_gs + ch] : 0.f;
        const float lo1 = (a.flags & VIDC_RELU1) ? 0.f : -INFINITY, lo2 = (a.flags & VIDC_RELU2) ? 0.f : -INFINITY;
        float* yb = a.y + (size_t)g * a.y_gs + ch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float out[4];
            at6(tc[r], out);
            const int oy = ty * 4 + r;
            if (oy >= a.H) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ox = tx * 4 + s;
                if (ox >= a.W) continue;
                float o = fmaxf(out[s] * s1 + b1, lo1);
                if (aff2) o = fmaxf(o * s2 + b2, lo2);
                yb[((size_t)(b * a.H + oy) * a.W + ox) * a.ldy] = o;
            }
        }
    }
}

// U [36][Cout][Cin] (vidc_winograd_weight_transform, m = 4) -> fragment order [36][Cout / 32][Cin / 32][nblk 2][h 2][lane 64][4]: lane = 16 kq + n holds
// U[pos][32 nb + 16 nblk + n][32 kc + 8 kq + 4 h + 0..3].  One thread per float4.
__global__ void __launch_bounds__(256)
wino4_pack_kernel(const float* __restrict__ u, float* __restrict__ o, int Cout, int Cin, long long total4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int NK = Cin / KC, nbn = Cout / NBW;
    const int lane = (int)(i & 63), h = (int)((i >> 6) & 1), nblk = (int)((i >> 7) & 1);
    long long r = i >> 8;
    const int kc = (int)(r % NK); r /= NK;
    const int nb = (int)(r % nbn);
    const int pos = (int)(r / nbn);
    const int n = lane & 15, kq = lane >> 4;
    const float4 v = *reinterpret_cast<const float4*>(u + ((size_t)pos * Cout + nb * NBW + nblk * 16 + n) * Cin + kc * KC + kq * 8 + h * 4);
    *reinterpret_cast<float4*>(o + i * 4) = v;
}

template <int DBG>
int launch_dbg(const FArgs& a, long long wgs, hipStream_t st) {
    static bool attr_set[64] = {};
    int dev = 0;
    VIDC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino4_fused_kernel<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(wino4_fused_kernel<DBG>, dim3((unsigned)wgs), dim3(NTHREADS), LDS_BYTES, st, a);
    VIDC_CHECK_LAUNCH("wino4_fused_kernel");
    return VIDC_OK;
}

}  // namespace

namespace vidc {

// Launch behind vidc_conv2d_bn_act for VIDC_TILE_WINO4_FUSED (csrc/conv_mfma.hip validates the common fields first): the descriptor is the 3x3 /
// stride 1 / pad 1 conv's, except that `w` holds U = G g G^T of vidc_winograd_weight_transform(m = 4) in the fragment order of vidc_winograd_weight_pack_fused, 36 Cout Cin floats per group (= w_gs).
int launch_wino4_fused(const vidc_conv_desc& d, hipStream_t st) {
    VIDC_REQUIRE(d.precision == VIDC_PREC_FP32, VIDC_ERR_SHAPE, "conv (fused Winograd): fp32 arithmetic only");
    VIDC_REQUIRE(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dilation <= 1, VIDC_ERR_SHAPE, "conv (fused Winograd): 3x3 / stride 1 / pad 1 / dense only");
    VIDC_REQUIRE(d.Cin % KC == 0 && d.Cout % NBW == 0 && d.ldx % 2 == 0, VIDC_ERR_SHAPE, "conv (fused Winograd): Cin %% 32 == 0, Cout %% 32 == 0, even ldx");
    VIDC_REQUIRE(!(d.flags & ~(VIDC_RELU1 | VIDC_AFFINE2 | VIDC_RELU2)), VIDC_ERR_SHAPE, "conv (fused Winograd): flags 0x%x not supported (affine / ReLU / second affine / ReLU only)", d.flags);
    VIDC_REQUIRE(d.w_gs == (long long)NPOS * d.Cout * d.Cin || d.groups == 1, VIDC_ERR_SHAPE, "conv (fused Winograd): w = U of vidc_winograd_weight_pack_fused, w_gs = 36 * Cout * Cin");
    VIDC_REQUIRE((reinterpret_cast<uintptr_t>(d.x) & 7) == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 && d.x_gs % 2 == 0, VIDC_ERR_SHAPE, "conv (fused Winograd): x 8-byte, U 16-byte aligned");
    FArgs a;
    a.x = d.x; a.u = d.w; a.y = d.y; a.s1 = d.scale1; a.b1 = d.shift1; a.s2 = d.scale2; a.b2 = d.shift2;
    a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.ldx = d.ldx; a.Cout = d.Cout; a.ldy = d.ldy;
    a.th = (d.H + 3) / 4; a.tw = (d.W + 3) / 4;
    a.T = d.B * a.th * a.tw;
    a.flags = d.flags; a.nbn = d.Cout / NBW; a.gn = d.groups * a.nbn;
    a.x_gs = d.x_gs; a.w_gs = d.w_gs; a.y_gs = d.y_gs; a.p_gs = d.p_gs;
    VIDC_REQUIRE((long long)d.B * d.H * d.W * d.ldx * 4 < (1ll << 30), VIDC_ERR_SHAPE, "conv (fused Winograd): the input must stay below 1 GiB (out-of-range taps are encoded in the offset)");
    a.x_bytes = (unsigned)((long long)d.B * d.H * d.W * d.ldx * 4);
    const long long wgs = (long long)((a.T + TB - 1) / TB) * a.gn;
    VIDC_REQUIRE(wgs < (1ll << 31), VIDC_ERR_SHAPE, "conv (fused Winograd): grid too large");
#ifdef VIDC_WFUSED_ATTRIB
    static const int dbg = [] { const char* e = getenv("VIDC_WFUSED_DBG"); return e ? atoi(e) : 0; }();
    switch (dbg) {
        case 1: return launch_dbg<1>(a, wgs, st);
        case 2: return launch_dbg<2>(a, wgs, st);
        case 4: return launch_dbg<4>(a, wgs, st);
        case 8: return launch_dbg<8>(a, wgs, st);
        case 3: return launch_dbg<3>(a, wgs, st);
        case 6: return launch_dbg<6>(a, wgs, st);
        case 7: return launch_dbg<7>(a, wgs, st);
        case 15: return launch_dbg<15>(a, wgs, st);
        default: break;
    }
#endif
    return launch_dbg<0>(a, wgs, st);
}

}  // namespace vidc

// U of vidc_winograd_weight_transform(m = 4), [36][Cout][Cin], into the fragment order the fused kernel streams (one group per call; same size).
extern "C" int vidc_winograd_weight_pack_fused(const float* u, float* u_packed, int Cout, int Cin, vidc_stream_t stream) {
    VIDC_REQUIRE(u && u_packed && u != u_packed, VIDC_ERR_NULL, "vidc_winograd_weight_pack_fused: null pointer (or in place)");
    VIDC_REQUIRE(Cout > 0 && Cin > 0 && Cout % NBW == 0 && Cin % KC == 0, VIDC_ERR_SHAPE, "vidc_winograd_weight_pack_fused: Cout and Cin must be multiples of 32");
    const long long total4 = (long long)NPOS * Cout * Cin / 4;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), u, u_packed, Cout, Cin, total4);
    VIDC_CHECK_LAUNCH("wino4_pack_kernel");
    return VIDC_OK;
}
