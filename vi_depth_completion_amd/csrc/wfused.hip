// Winograd F(4x4, 3x3) in ONE launch for the small maps: input transform, the 36 transform-domain products and the output transform with the conv's
// epilogue, without the V and M tensors (VERDICT r5 items 2 and 3 iii).  Parity-green; 7 % faster than the three launches alone, +0.7 % on the stream
// (DESIGN 4.2); the engine records it where the measured table says 5 (or behind VIDC_WINO_FUSED), the committed table does not yet.
//
// On a 16x20 map (ResNet-101 layer 3 at the program batch bench.py times: 4 frames x 4 x 5 tiles = 80 tiles per pyramid, Cin = Cout = 256,
// networks/surface_normal.py:27-50) the three launches of csrc/winograd.hip + conv_mfma.hip take 7 + 28.5 + 9.5 us, of which three launch floors and the
// lock-step stage skeleton of a few-row GEMM are more than half: the products are 1.5 GFLOP = 9.6 us of the chip's fp32 MFMA rate.  Here a
// workgroup owns 16 tiles x 32 output channels of one group and ALL 36 positions (16 waves, 146 KB of LDS, one workgroup per CU), K in chunks of 16
// channels, one workgroup barrier per chunk:
//
//   patch stage: the 16 tiles' 6 x 6 patches of chunk it + 3 arrive by LDS-DMA -- 37 instructions of 1 KiB (16 pixel slots x 4 channel quads) spread over
//       the eight transform waves, no registers, pixels outside the map arrive as zeros (out-of-range offsets) -- into one of two stage buffers.
//   waves 8..15 (transform), two groups of four, thread = (tile, channel): group it & 1 reads its patch of chunk it + 2 from the stage and computes
//       V = B^T d B in registers (the formulas of csrc/winograd.hip, no contraction) while the OTHER group writes the V of chunk it + 1 -- computed one
//       iteration earlier -- as [pos][tile][16 ch] into the V buffer the product waves are not reading.  Every group has two product phases per chunk;
//       a transform wave never holds up a barrier.
//   waves 0..7 (products): wave w < 4 owns positions 5 w .. 5 w + 4, wave w >= 4 positions 20 + 4 (w - 4) ..: waves w and w + 4 share a SIMD = 9
//       positions per SIMD, and one of the two issues MFMAs while the other is held up issuing a load (a VMEM instruction costs its wave ~75-140 clk
//       whatever its width).  Per position 2 accumulators of v_mfma_f32_16x16x4_f32 (16 tiles x 16 channels each) over all of K.  A = V from LDS (one
//       ds_read_b128 per position and chunk, rows XOR-swizzled: conflict-free without padding); B = U straight from L2 into registers in FRAGMENT ORDER
//       (vidc_winograd_weight_pack_fused: every load instruction is 1 KiB contiguous; MFMA k-step ks of lane group kq multiplies channel 4 kq + ks of
//       the chunk -- the K order is free as long as A and B agree), one whole chunk ahead, across chunk boundaries and barriers.
//   epilogue: the 36 x 16 x 32 products go through LDS once, thread = (tile, channel) folds them with A^T (.) A, applies the conv's affine / ReLU
//       [/ second affine / ReLU] and stores its 4 x 4 pixels (128 contiguous bytes per pixel and workgroup).
//
// Workgroup L = (tile block, group, channel block) with the (group, channel block) pair fastest: the tile blocks that stream the same slice of U
// share L % 8 = one XCD's L2.  Result bits of a tile depend on nothing but the tile (fixed K order, fixed fold): a restriction to fewer groups
// (engine.Program.group_variant) or another batch leaves an item's bits alone (tests/test_wfused.py).
//
// Measured (tools/wfused_bench.py, weights HBM-cold, profiles/r6_wfused_bench.txt): layer 3 (4 groups) 39.9-42.4 us against 44.1-45.0 us for the three
// launches (37.5 against 44.3 inside the frame program); one group alone 26.0 / 27.2 and three groups 29.8 / 36.0 on the NB = 1 instantiation (16 output
// channels per workgroup for launches of few groups; with 32: 33.6 / 37.4); layer 2 44.8 / 39.5.  Attribution
// builds (make wfused_attrib, tools/wfused_attrib.sh): empty skeleton 10.7 us (= the launch floor), + MFMAs 29.4 (18.7 us of products: 160 workgroups use
// 160 of 256 CUs and the 16x16 MFMA form reads twice the operand bytes per FLOP of the 32x32 form), + U loads 33.5, + transform 41.8 (of which the fold 4);
// not waiting for the DMA changes nothing.  Earlier forms (git history): 56 -> 53 -> 46 us.
#include "common.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 16, NBW = 32, KC = 16, NPOS = 36;
constexpr int NPW = 8, NTW = 8, NTHREADS = 64 * (NPW + NTW);      // 8 product waves (5 or 4 positions each: 9 per SIMD), 2 groups of 4 transform waves
constexpr int VBUF = NPOS * TB * KC;                 // floats of one V buffer: [pos][tile][16 ch]
constexpr int XSLOT = 37;                            // pixel slots per tile in the patch stage (36 + 1 unused: the four tiles of a wave start on different banks)
constexpr int XBUF = TB * XSLOT * KC;                // floats of one patch-stage buffer: [tile][slot][16 ch]
constexpr int NDMA = TB * XSLOT / 16;                // LDS-DMA instructions per chunk: 16 pixel slots x 4 quads of channels each (1 KiB)
constexpr int MLD = 36;                              // row stride of the product rows in the epilogue (floats): 4 kq row groups -> 4 bank groups
constexpr size_t LDS_BYTES = (size_t)(2 * VBUF + 2 * XBUF) * sizeof(float);
static_assert(TB * XSLOT % 16 == 0, "whole DMA instructions");
static_assert((size_t)NPOS * TB * MLD * sizeof(float) <= LDS_BYTES, "the epilogue's product rows reuse the buffers");
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

struct FArgs {
    const float* x; const float* u; float* y; const float* s1; const float* b1; const float* s2; const float* b2;
    int H, W, Cin, ldx, Cout, ldy, th, tw, T, flags, nbn, gn;      // T = B * th * tw tiles; nbn = Cout / (16 NB) channel blocks of a workgroup; gn = groups * nbn
    long long x_gs, w_gs, y_gs, p_gs;
    unsigned x_bytes;
};

typedef __attribute__((address_space(3))) void lds_void_t;

// B^T d for one column / A^T m for one column: Lavin & Gray's F(4x4, 3x3) matrices, the operation order of csrc/winograd.hip Wino<4>
// (in two halves: rows 0..2 / 3..5 of B^T d share no intermediate)
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
template <typename V> __device__ __forceinline__ void bt6(const V (&d)[6], V (&t)[6]) {
    V lo[3], hi[3];
    bt6_lo(d, lo);
    bt6_hi(d, hi);
#pragma unroll
    for (int i = 0; i < 3; ++i) { t[i] = lo[i]; t[3 + i] = hi[i]; }
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
// LDS accesses of the transform waves go through inline asm: hipcc cannot prove that an ordinary LDS access does not alias an in-flight LDS-DMA and
// would drain it (s_waitcnt vmcnt(0)) in front of every read
template <int OFF> __device__ __forceinline__ float lds_read_b32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ void lds_write_b32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }

// One product wave: positions pos0 .. pos0 + PW - 1, all of K, 16 tiles x NB blocks of 16 output channels.
template <int PW, int NB, int DBG>
__device__ __forceinline__ void product_wave(const FArgs& a, float* smem, unsigned lds0, int lane, int pos0, int g, int nb, int NK) {
    const int n = lane & 15, kq = lane >> 4;
    // U in fragment order (vidc_winograd_weight_pack_fused): [g][pos][nb][kc][nblk][lane][4] -- one load instruction = 1 KiB contiguous
    // (packed per 32 channels: [pos][Cout / 32][kc][nblk 2][lane][4]; a workgroup of one 16-channel block takes nblk = nb & 1 of pair nb >> 1)
    const int n32 = a.Cout / 32, pair = NB == 2 ? nb : nb >> 1, half = NB == 2 ? 0 : nb & 1;
    const float* ub = a.u + (size_t)g * a.w_gs + ((size_t)pos0 * n32 + pair) * ((size_t)NK * 512) + half * 256 + lane * 4;
    const size_t pos_stride = (size_t)n32 * NK * 512;
    auto load_b = [&](int p, int kc, f32x4 (&dst)[NB]) {
        const float* s = ub + (size_t)p * pos_stride + (size_t)kc * 512;
        dst[0] = *reinterpret_cast<const f32x4*>(s);
        if constexpr (NB == 2) dst[1] = *reinterpret_cast<const f32x4*>(s + 256);
    };
    f32x4 acc[PW][NB];
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bq[PW][NB];                                  // a whole chunk of the wave's U fragments ahead
#pragma unroll
    for (int p = 0; p < PW; ++p) load_b(p, 0, bq[p]);
    // A fragment of (position, tile m = lane & 15): channels 4 kq .. 4 kq + 3 of the 64-byte row = 16-byte unit kq XOR (m >> 2) & 3
    const unsigned a_off = (unsigned)((n * KC + ((kq ^ ((n >> 2) & 3)) * 4)) * 4);
    const unsigned a_wave = lds0 + (unsigned)(pos0 * TB * KC * 4);

    __builtin_amdgcn_s_barrier();                     // the three barriers of the transform waves' prologue
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                     // chunk 0 of V is in buffer 0
    for (int kc = 0; kc < NK; ++kc) {
        const unsigned vb = a_wave + (unsigned)((kc & 1) * VBUF * 4);
        const int kn = kc + 1 < NK ? kc + 1 : kc;          // (the look-ahead loads of the last chunk re-read it: harmless, no tail case)
        f32x4 fa[2];
        fa[0] = lds_read_b128(vb + a_off);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int cur = p & 1, nxt = cur ^ 1;
            if (p + 1 < PW) {
                fa[nxt] = lds_read_b128(vb + (unsigned)((p + 1) * TB * KC * 4) + a_off);
                wait_lgkmcnt<1>();
            } else {
                wait_lgkmcnt<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if constexpr (!(DBG & 1)) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][ks], bq[p][j][ks], acc[p][j], 0, 0, 0);
                    else if (ks == 0) acc[p][j][0] += fa[cur][0] + fa[cur][3] + bq[p][j][0] + bq[p][j][3];
            if constexpr (!(DBG & 2)) load_b(p, kn, bq[p]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();                 // every wave has read buffer kc & 1; chunk kc + 1 is in the other one
    }
    // products -> LDS rows [pos][tile][MLD]: lane holds rows 4 kq + i, column n of each 16 x 16 block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* mx = smem;
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx[((pos0 + p) * TB + 4 * kq + i) * MLD + j * 16 + n] = acc[p][j][i];
}

// DBG (attribution builds, -DVIDC_WFUSED_ATTRIB + VIDC_WFUSED_DBG): 1 no MFMAs, 2 no U loads in the loop, 4 no transform (barriers only), 8 no fold / stores
// NB = 16-channel blocks per workgroup: 2 (32 output channels) when the layer fills the chip that way, 1 for launches of few groups (the head / tail ticks of
// a stream: twice the workgroups, half the products each).  A tile's result bits do not depend on NB: the same K order and fold per output element.
template <int NB, int DBG>
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
        if (w < 4) product_wave<5, NB, DBG>(a, smem, lds0, tid & 63, 5 * w, g, nb, NK);
        else product_wave<4, NB, DBG>(a, smem, lds0, tid & 63, 20 + 4 * (w - 4), g, nb, NK);
    } else {
        // ------------------------------------------------------------------ transform waves: two groups of four, thread = (tile m, channel c of the chunk)
        const int tt = tid - 64 * NPW, lane = tid & 63;
        const int w8 = __builtin_amdgcn_readfirstlane(tt >> 6), grp = w8 >> 2;
        const int u_ = tt & 255, m = u_ >> 4, c = u_ & 15;
        const int tpf = a.th * a.tw;
        // ---- patch stage: LDS-DMA instruction j of a chunk moves pixel slots 16 j .. 16 j + 15 (slot = 37 tile + pixel) x 4 channel quads, 1 KiB, into
        // [slot][16 ch]; a pixel outside the map, the 37th slot of a tile and tiles beyond T get an out-of-range offset and arrive as zeros.
        // Wave w8 issues instructions w8, w8 + 8, ...; their per-lane offsets are computed once.
        constexpr unsigned OOB = 0x80000000u;
        unsigned voff[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = w8 + 8 * i;
            const int slot = 16 * j + (lane >> 2), quad = lane & 3;
            const int ts = slot / XSLOT, px = slot - ts * XSLOT;
            const int t = tb * TB + ts;
            bool ok = j < NDMA && px < 36 && t < a.T;
            const int tc = ok ? t : 0;
            const int b = tc / tpf, rem = tc - b * tpf, ty = rem / a.tw, tx = rem - ty * a.tw;
            const int r = px / 6, s = px - r * 6;
            const int iy = ty * 4 - 1 + r, ix = tx * 4 - 1 + s;
            ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            voff[i] = ok ? (unsigned)(((size_t)((b * a.H + iy) * a.W + ix) * a.ldx + (size_t)g * a.x_gs + quad * 4) * 4) : OOB;
        }
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
        auto dma = [&](int cc) {                      // chunk cc -> stage buffer cc & 1
            float* dst = smem + 2 * VBUF + (cc & 1) * XBUF;
            const int soff = cc * KC * 4;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (w8 + 8 * i < NDMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_t*)(dst + (w8 + 8 * i) * 256), 16, (int)voff[i], soff, 0, 0);
        };
        // ---- phase A: the thread's 6 x 6 patch of chunk cc from the stage -> V = B^T d B in registers (out[36]); phase B (one iteration later, when the
        // product waves have left that V buffer): out -> [pos][tile][16 ch], 16-byte unit (c >> 2) XOR (m >> 2) & 3
        float out[36];
        const unsigned x_thr = lds0 + (unsigned)((2 * VBUF + (m * XSLOT) * KC + c) * 4);
        const unsigned v_thr = lds0 + (unsigned)((m * KC + (((c >> 2) ^ ((m >> 2) & 3)) * 4) + (c & 3)) * 4);
        auto phase_a = [&](int cc) {
            const unsigned xa = x_thr + (unsigned)((cc & 1) * XBUF * 4);
            float D[36];
#define VIDC_RD(i) D[i] = lds_read_b32<(i) * KC * 4>(xa);
            VIDC_RD(0) VIDC_RD(1) VIDC_RD(2) VIDC_RD(3) VIDC_RD(4) VIDC_RD(5) VIDC_RD(6) VIDC_RD(7) VIDC_RD(8) VIDC_RD(9) VIDC_RD(10) VIDC_RD(11)
            VIDC_RD(12) VIDC_RD(13) VIDC_RD(14) VIDC_RD(15) VIDC_RD(16) VIDC_RD(17) VIDC_RD(18) VIDC_RD(19) VIDC_RD(20) VIDC_RD(21) VIDC_RD(22) VIDC_RD(23)
            VIDC_RD(24) VIDC_RD(25) VIDC_RD(26) VIDC_RD(27) VIDC_RD(28) VIDC_RD(29) VIDC_RD(30) VIDC_RD(31) VIDC_RD(32) VIDC_RD(33) VIDC_RD(34) VIDC_RD(35)
#undef VIDC_RD
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3]), "+v"(D[4]), "+v"(D[5]), "+v"(D[6]), "+v"(D[7]), "+v"(D[8]), "+v"(D[9]), "+v"(D[10]), "+v"(D[11]), "+v"(D[12]), "+v"(D[13]), "+v"(D[14]), "+v"(D[15]), "+v"(D[16]), "+v"(D[17]) :: "memory");
            asm volatile("" : "+v"(D[18]), "+v"(D[19]), "+v"(D[20]), "+v"(D[21]), "+v"(D[22]), "+v"(D[23]), "+v"(D[24]), "+v"(D[25]), "+v"(D[26]), "+v"(D[27]), "+v"(D[28]), "+v"(D[29]), "+v"(D[30]), "+v"(D[31]), "+v"(D[32]), "+v"(D[33]), "+v"(D[34]), "+v"(D[35]) :: "memory");
            float tcol[6][6];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                float col[6], o6[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) col[r] = D[r * 6 + s];
                bt6(col, o6);
#pragma unroll
                for (int r = 0; r < 6; ++r) tcol[r][s] = o6[r];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float o6[6];
                bt6(tcol[r], o6);
#pragma unroll
                for (int s = 0; s < 6; ++s) out[r * 6 + s] = o6[s];
            }
        };
        auto phase_b = [&](int cc) {
            const unsigned va = v_thr + (unsigned)((cc & 1) * VBUF * 4);
#define VIDC_WR(i) lds_write_b32<(i) * TB * KC * 4>(va, out[i]);
            VIDC_WR(0) VIDC_WR(1) VIDC_WR(2) VIDC_WR(3) VIDC_WR(4) VIDC_WR(5) VIDC_WR(6) VIDC_WR(7) VIDC_WR(8) VIDC_WR(9) VIDC_WR(10) VIDC_WR(11)
            VIDC_WR(12) VIDC_WR(13) VIDC_WR(14) VIDC_WR(15) VIDC_WR(16) VIDC_WR(17) VIDC_WR(18) VIDC_WR(19) VIDC_WR(20) VIDC_WR(21) VIDC_WR(22) VIDC_WR(23)
            VIDC_WR(24) VIDC_WR(25) VIDC_WR(26) VIDC_WR(27) VIDC_WR(28) VIDC_WR(29) VIDC_WR(30) VIDC_WR(31) VIDC_WR(32) VIDC_WR(33) VIDC_WR(34) VIDC_WR(35)
#undef VIDC_WR
        };
        auto fence_barrier = [&]() {                  // own DMA landed, own LDS writes done
            if constexpr (DBG & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (attribution: the DMA is not waited for -- wrong results, right schedule)
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        // ---- prologue: chunks 0 and 1 staged; A(0) by group 0, A(1) by group 1; chunk 2 staged; B(0)
        dma(0);
        if (NK > 1) dma(1);
        fence_barrier();
        if (grp == 0) phase_a(0);
        else if (NK > 1) phase_a(1);
        fence_barrier();
        if (NK > 2) dma(2);
        if (grp == 0) phase_b(0);
        fence_barrier();
        // ---- iteration it (the product waves multiply chunk it): group it & 1 does A(it + 2), the other group B(it + 1); chunk it + 3 is staged
        for (int it = 0; it < NK; ++it) {
            if (!(DBG & 4)) {
                if (it + 3 < NK) dma(it + 3);
                if ((it & 1) == grp) {
                    if (it + 2 < NK) phase_a(it + 2);
                } else {
                    if (it + 1 < NK) phase_b(it + 1);
                }
            }
            fence_barrier();
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------------- fold + epilogue: thread = (tile m, channel c)
    if (tid < TB * 16 * NB) {
        const int m = tid / (16 * NB), c = tid - m * (16 * NB);
        const int t = tb * TB + m;
        if (t >= a.T || (DBG & 8)) return;
        const int tpf = a.th * a.tw;
        const int b = t / tpf, rem = t - b * tpf, ty = rem / a.tw, tx = rem - ty * a.tw;
        const float* mx = smem + m * MLD + c;
        float tc[4][6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float col[6], o4[4];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = mx[(r * 6 + s) * TB * MLD];
            at6(col, o4);
#pragma unroll
            for (int r = 0; r < 4; ++r) tc[r][s] = o4[r];
        }
        const int ch = nb * 16 * NB + c;
        const float s1 = a.s1[(size_t)g * a.p_gs + ch], b1 = a.b1[(size_t)g * a.p_gs + ch];
        const bool aff2 = a.flags & VIDC_AFFINE2;
        const float s2 = aff2 ? a.s2[(size_t)g * a.p_gs + ch] : 0.f, b2 = aff2 ? a.b2[(size_t)g * a.p_gs + ch] : 0.f;
        const float lo1 = (a.flags & VIDC_RELU1) ? 0.f : -INFINITY, lo2 = (a.flags & VIDC_RELU2) ? 0.f : -INFINITY;
        float* yb = a.y + (size_t)g * a.y_gs + ch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o4[4];
            at6(tc[r], o4);
            const int oy = ty * 4 + r;
            if (oy >= a.H) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ox = tx * 4 + s;
                if (ox >= a.W) continue;
                float o = fmaxf(o4[s] * s1 + b1, lo1);
                if (aff2) o = fmaxf(o * s2 + b2, lo2);
                yb[((size_t)(b * a.H + oy) * a.W + ox) * a.ldy] = o;
            }
        }
    }
}

// U [36][Cout][Cin] (vidc_winograd_weight_transform, m = 4) -> fragment order [36][Cout / 32][Cin / 16][nblk 2][lane 64][4]: lane = 16 kq + n holds
// U[pos][32 nb + 16 nblk + n][16 kc + 4 kq + 0..3].  One thread per float4.
__global__ void __launch_bounds__(256)
wino4_pack_kernel(const float* __restrict__ u, float* __restrict__ o, int Cout, int Cin, long long total4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int NK = Cin / KC, nbn = Cout / NBW;
    const int lane = (int)(i & 63), nblk = (int)((i >> 6) & 1);
    long long r = i >> 7;
    const int kc = (int)(r % NK); r /= NK;
    const int nb = (int)(r % nbn);
    const int pos = (int)(r / nbn);
    const int n = lane & 15, kq = lane >> 4;
    const float4 v = *reinterpret_cast<const float4*>(u + ((size_t)pos * Cout + nb * NBW + nblk * 16 + n) * Cin + kc * KC + kq * 4);
    *reinterpret_cast<float4*>(o + i * 4) = v;
}

template <int NB, int DBG>
int launch_nb(const FArgs& a, long long wgs, hipStream_t st) {
    static bool attr_set[64] = {};
    int dev = 0;
    VIDC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino4_fused_kernel<NB, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((wino4_fused_kernel<NB, DBG>), dim3((unsigned)wgs), dim3(NTHREADS), LDS_BYTES, st, a);
    VIDC_CHECK_LAUNCH("wino4_fused_kernel");
    return VIDC_OK;
}

template <int DBG>
int launch_dbg(const FArgs& a, long long wgs, hipStream_t st) { return a.nbn * 32 == a.Cout ? launch_nb<2, DBG>(a, wgs, st) : launch_nb<1, DBG>(a, wgs, st); }

}  // namespace

namespace vidc {

// Launch behind vidc_conv2d_bn_act for VIDC_TILE_WINO4_FUSED (csrc/conv_mfma.hip validates the common fields first): the descriptor is the 3x3 /
// stride 1 / pad 1 conv's, except that `w` holds U = G g G^T of vidc_winograd_weight_transform(m = 4) in the fragment order of vidc_winograd_weight_pack_fused, 36 Cout Cin floats per group (= w_gs).
int launch_wino4_fused(const vidc_conv_desc& d, hipStream_t st) {
    VIDC_REQUIRE(d.precision == VIDC_PREC_FP32, VIDC_ERR_SHAPE, "conv (fused Winograd): fp32 arithmetic only");
    VIDC_REQUIRE(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dilation <= 1, VIDC_ERR_SHAPE, "conv (fused Winograd): 3x3 / stride 1 / pad 1 / dense only");
    VIDC_REQUIRE(d.Cin % KC == 0 && d.Cout % NBW == 0 && d.ldx % 4 == 0, VIDC_ERR_SHAPE, "conv (fused Winograd): Cin %% 16 == 0, Cout %% 32 == 0, ldx %% 4 == 0");
    VIDC_REQUIRE(!(d.flags & ~(VIDC_RELU1 | VIDC_AFFINE2 | VIDC_RELU2)), VIDC_ERR_SHAPE, "conv (fused Winograd): flags 0x%x not supported (affine / ReLU / second affine / ReLU only)", d.flags);
    VIDC_REQUIRE(d.w_gs == (long long)NPOS * d.Cout * d.Cin || d.groups == 1, VIDC_ERR_SHAPE, "conv (fused Winograd): w = U of vidc_winograd_weight_pack_fused, w_gs = 36 * Cout * Cin");
    VIDC_REQUIRE((reinterpret_cast<uintptr_t>(d.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 && d.x_gs % 4 == 0, VIDC_ERR_SHAPE, "conv (fused Winograd): x and U 16-byte aligned");
    FArgs a;
    a.x = d.x; a.u = d.w; a.y = d.y; a.s1 = d.scale1; a.b1 = d.shift1; a.s2 = d.scale2; a.b2 = d.shift2;
    a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.ldx = d.ldx; a.Cout = d.Cout; a.ldy = d.ldy;
    a.th = (d.H + 3) / 4; a.tw = (d.W + 3) / 4;
    a.T = d.B * a.th * a.tw;
    // 32 output channels per workgroup when that gives more than half a chip of workgroups, else 16 (VIDC_WFUSED_NB=1 / 2 forces: A-B runs)
    static const int force_nb = [] { const char* e = getenv("VIDC_WFUSED_NB"); return e ? atoi(e) : 0; }();
    const long long wgs32 = (long long)((d.B * a.th * a.tw + TB - 1) / TB) * d.groups * (d.Cout / NBW);
    const int nbk = force_nb == 1 || force_nb == 2 ? force_nb : (wgs32 * 2 <= 256 ? 1 : 2);
    a.flags = d.flags; a.nbn = d.Cout / (16 * nbk); a.gn = d.groups * a.nbn;
    a.x_gs = d.x_gs; a.w_gs = d.w_gs; a.y_gs = d.y_gs; a.p_gs = d.p_gs;
    VIDC_REQUIRE((long long)d.B * d.H * d.W * d.ldx * 4 < (1ll << 30), VIDC_ERR_SHAPE, "conv (fused Winograd): the input must stay below 1 GiB");
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
        case 16: return launch_dbg<16>(a, wgs, st);
        default: break;
    }
#endif
    return launch_dbg<0>(a, wgs, st);
}

}  // namespace vidc

// U of vidc_winograd_weight_transform(m = 4), [36][Cout][Cin], into the fragment order the fused kernel streams (one group per call; same size).
extern "C" int vidc_winograd_weight_pack_fused(const float* u, float* u_packed, int Cout, int Cin, vidc_stream_t stream) {
    VIDC_REQUIRE(u && u_packed && u != u_packed, VIDC_ERR_NULL, "vidc_winograd_weight_pack_fused: null pointer (or in place)");
    VIDC_REQUIRE(Cout > 0 && Cin > 0 && Cout % NBW == 0 && Cin % KC == 0, VIDC_ERR_SHAPE, "vidc_winograd_weight_pack_fused: Cout must be a multiple of 32, Cin of 16");
    const long long total4 = (long long)NPOS * Cout * Cin / 4;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), u, u_packed, Cout, Cin, total4);
    VIDC_CHECK_LAUNCH("wino4_pack_kernel");
    return VIDC_OK;
}
