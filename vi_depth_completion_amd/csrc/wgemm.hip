// Streamed grouped GEMM for the Winograd-domain products with FEW rows (round 6; VERDICT r5 item 3 i).
//
// A Winograd layer on the small maps is a^2 x G independent GEMMs  M[gi] (tiles x Cout) = V[gi] (tiles x Cin) * U[gi]^T  with only 80 rows at the
// program batch bench.py times (4 frames x 4 x 5 tiles of a 16x20 map): ResNet-101 layer 3 (22 launches per tick of 144 groups, K = N = 256), layer 4 and
// the two deepest decoder levels -- 1.19 of the 7.16 ms of the fp32 tick at 36-72 % of the MFMA peak.  Through conv_igemm_f32 every (group, n-tile,
// m-tile) is a workgroup of its own that lives for 4 pipeline stages: 1728 workgroups per launch, each paying kernel-argument loads, address decode,
// a cold ring fill and an epilogue for 1.7 us of MFMA work (profiles/r5_conv_phase_profile_M1280.txt: 3-7 k clk of set-up per workgroup).
//
// Here ONE workgroup owns an n-tile of 32 output channels and a CHUNK of consecutive groups and streams them through one continuous LDS-DMA ring:
// the loads of group gi + 1 are in flight while group gi is multiplied, its accumulator is reduced over the two k-slices, written and the next
// one starts -- one prologue per workgroup, not per group.  Same data path as conv_mfma.hip (raw_buffer_load ... lds, 16 B per lane, XOR-swizzled
// 128-byte rows, counted vmcnt, inline-asm ds_read_b128 fragments, v_mfma_f32_32x32x2_f32 -- exact fp32 products and sums).
//
//   workgroup = 3 m-blocks (96 rows: M <= 96) x NB n-blocks x 2 k-slices of waves; wave (kq, mb, nb) owns one 32 x 32 block over the K units u = 2 st + kq.
//   ring: NSR slots of [kq][96 A rows | 32 NB B rows][32 floats]; + the k-reduction scratch (4 KB per block) + a 1 KB DMA sink.
//     <NB 1, NSR 2>:  6 waves,  77 KB -> two workgroups per CU, one stage of look-ahead each;
//     <NB 2, NSR 3>: 12 waves, 145 KB -> one workgroup per CU, three waves per SIMD, TWO stages (80 KB) in flight: what the measurements of
//                    round 6 asked for -- the general tiles and <1, 2> alike take ~29 us for the layer-3 product whether its weights are
//                    HBM-cold or L2-warm (26-32 us, profiles/r5_weight_warmth_b4.txt): with one stage of look-ahead every stage costs an L2 / HBM
//                    round trip (~2 us) instead of its 0.4-1.3 us of MFMA work.
//   work list: chunk c of the `chunks` contiguous group ranges (the first groups % chunks ones hold one group more and are dispatched first) x n-tile;
//   workgroup w = (c, t) with c = (w / (8 TN)) * 8 + (w & 7), t = (w >> 3) % TN: the TN n-tiles of a chunk share its V rows and sit on ONE XCD.
//   Result bits do not depend on chunks / groups in the launch: a group's K order and the k-slice sum (slice 0 + slice 1) are fixed.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int BM = 96, BK = 32, WK = 2, MB = BM / 32;
constexpr int SINK = 256;

struct WArgs {
    const float* a; const float* b; float* c; const float* scale; const float* shift;
    int M, N, K, lda, ldc, groups, chunks, tiles_n, relu, dbg;      // dbg (VIDC_WGEMM_DBG, attribution runs only): 2 B zero-sourced, 4 A zero-sourced, 8 no stores (1 = no MFMAs: builds with -DVIDC_WGEMM_ATTRIB)
    long long a_gs, b_gs, c_gs;
    unsigned a_bytes, b_bytes;
};

template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int N_> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory"); }
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

template <int NB, int NSR> struct Cfg {
    static constexpr int BN = 32 * NB, NWV = MB * NB * WK;
    static constexpr int ROWS = BM + BN;                  // rows of one k-slice of a stage
    static constexpr int STAGE = ROWS * BK * WK;          // floats per ring slot
    static constexpr int LA = 24 / NWV;                   // A instructions (8 rows x 128 B) per wave and stage: 24 = 2 k-slices x 12
    static constexpr int LB = (8 * NB + NWV - 1) / NWV;   // B instructions per wave and stage (the last ones of some waves go to the sink)
    static constexpr int LPS = LA + LB;
    static constexpr int SCRATCH = MB * NB * 1024;        // k-slice 1's accumulator blocks
    static constexpr size_t LDS_BYTES = (size_t)(NSR * STAGE + SCRATCH + SINK) * sizeof(float);
    static_assert(24 % NWV == 0 && NSR >= 2 && NSR <= 4, "wave count must divide the 24 A instructions of a stage");
};

template <int NB, int NSR>
__global__ void __launch_bounds__(64 * MB * NB * WK)
wgemm_stream_kernel(const WArgs a) {
    typedef Cfg<NB, NSR> C_;
    constexpr int BN = C_::BN, NWV = C_::NWV, ROWS = C_::ROWS, STAGE = C_::STAGE, LA = C_::LA, LB = C_::LB, LPS = C_::LPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave / (MB * NB), wq = wave - kq * (MB * NB), mb = wq / NB, nb = wq - mb * NB;
    // ---- work item ------------------------------------------------------------------------------------------------------------------------
    const int w = blockIdx.x, TN = a.tiles_n;
    const int chunk = (w / (8 * TN)) * 8 + (w & 7), tn = (w >> 3) % TN;
    if (chunk >= a.chunks) return;                    // (the chunk count is padded to a multiple of 8 by the grid)
    const int base = a.groups / a.chunks, rem = a.groups - base * a.chunks;
    const int g0 = chunk * base + min(chunk, rem), ng = base + (chunk < rem ? 1 : 0);
    if (ng == 0) return;
    const int n0 = tn * BN;
    const int KS = a.K / (BK * WK);                   // stages per group
    const int nst = ng * KS;

    // ---- DMA sources: one descriptor per operand, 32-bit byte offsets, out-of-range offsets read zeros -------------------------------------
    constexpr unsigned OOB = 0x80000000u;
    auto uniform_ptr = [](const float* p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.a), 0, (int)a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.b), 0, (int)a.b_bytes, 0x00020000);
    const int lrow = lane >> 3, lc = lane & 7;
    // A: instruction qa = wave + NWV j (j < LA) of the 24 = 2 k-slices x 12 eight-row groups; B: qb = wave + NWV j (j < LB) of the 8 NB (beyond: the sink)
    unsigned off[LPS];          // per-lane byte offset of (row, swizzled 16-byte chunk) inside group 0, K unit 0
    int dst[LPS], smul[LPS];    // float offset of the instruction's 1 KiB in slot 0 and the slot stride (wave-uniform); the sink: beyond the ring, stride 0
    int kslice[LPS];            // the k-slice (K unit parity) the instruction feeds
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int qa = wave + NWV * j, ks = qa / 12, r8 = qa - ks * 12;
        const int row = r8 * 8 + lrow;
        const int csw = (lc ^ ((row >> 1) & 7)) * 4;
        off[j] = row < a.M ? (unsigned)((row * a.lda + csw) * 4) : OOB;
        dst[j] = (ks * ROWS + r8 * 8) * BK;
        smul[j] = STAGE;
        kslice[j] = ks;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int qb = wave + NWV * j;
        const bool real = qb < 8 * NB;
        const int ks = real ? qb / (4 * NB) : 0, r8 = real ? qb - ks * (4 * NB) : 0;
        const int row = r8 * 8 + lrow;                 // row of the n-tile; 96 + row in the slot: (96 + row) >> 1 & 7 == row >> 1 & 7
        const int csw = (lc ^ ((row >> 1) & 7)) * 4;
        off[LA + j] = real ? (unsigned)(((n0 + row) * a.K + csw) * 4) : OOB;
        dst[LA + j] = real ? (ks * ROWS + BM + r8 * 8) * BK : NSR * STAGE + C_::SCRATCH;
        smul[LA + j] = real ? STAGE : 0;
        kslice[LA + j] = ks;
    }
    // stage st (0 .. nst-1) = K units 2 (st % KS) + {0, 1} of group g0 + st / KS; stages beyond nst are issued zero-sourced, so that every wave has
    // exactly LPS loads per stage in flight and the counted waits below need no tail cases
    int i_gi = 0, i_ku = 0;                           // (group, K stage) of the next stage to issue: advanced by issue_begin(), no division in the stream
    long long is_g = 0;                               // the stage being issued: its group, K stage, liveness and ring slot
    int is_ku = 0, is_slot = 0;
    bool is_live = false;
    auto issue_begin = [&](int st, int slot) {
        is_ku = i_ku;
        is_live = st < nst;
        is_g = g0 + i_gi;
        is_slot = slot;
        if (++i_ku == KS) { i_ku = 0; ++i_gi; }
    };
    // one of the stage's LPS instructions; the main loop places them BETWEEN its MFMA groups: issued in one burst at the top of a stage the twelve
    // waves' DMA instructions queue up in the CU's one address path while the matrix pipes wait (attribution runs: loads + skeleton and MFMAs added up)
    auto issue_one = [&](int j) {
        const unsigned add = (unsigned)(((j < LA ? is_g * a.a_gs : is_g * a.b_gs) + (long long)(is_ku * WK + kslice[j]) * BK) * 4);
        const unsigned voff = (is_live && !(a.dbg & (j < LA ? 4 : 2))) ? off[j] + add : OOB;      // (OOB + add stays >= 2^31: add < 2^31 by validate())
        float* d = smem + dst[j] + is_slot * smul[j];
        if (j < LA) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void_t*)d, 16, (int)voff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void_t*)d, 16, (int)voff, 0, 0, 0);
    };
    auto issue = [&](int st, int slot) {
        issue_begin(st, slot);
#pragma unroll
        for (int j = 0; j < LPS; ++j) issue_one(j);
    };
    // epilogue operands (identity for the Winograd products; one affine + ReLU supported), fetched before the first DMA so that the counted
    // waits below see DMA loads only
    const int li = lane & 31, lh = lane >> 5;
    float e_s = a.scale[n0 + nb * 32 + li], e_b = a.shift[n0 + nb * 32 + li];
    const float lo1 = a.relu ? 0.f : -INFINITY;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_s), "+v"(e_b)::"memory");      // (the two loads are complete HERE: no compiler-placed vmcnt wait inside the stream)

#pragma unroll
    for (int s = 0; s < NSR - 1; ++s) issue(s, s);    // prologue: NSR - 1 stages ahead

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) float*)smem);
    const unsigned a_base = lds0 + 4u * (unsigned)((kq * ROWS + mb * 32 + li) * BK);
    const unsigned b_base = lds0 + 4u * (unsigned)((kq * ROWS + BM + nb * 32 + li) * BK);
    const int sw = (li >> 1) & 7;
    unsigned coff[BK / 8];
#pragma unroll
    for (int sub = 0; sub < BK / 8; ++sub) coff[sub] = (unsigned)(((sub * 2 + lh) ^ sw) * 16);
    // k-reduction scratch, touched with inline asm like the fragments: hipcc cannot prove that an ordinary LDS access does not alias an in-flight
    // LDS-DMA and would drain the ring (s_waitcnt vmcnt(0)) in front of it
    const unsigned red0 = lds0 + 4u * (unsigned)(NSR * STAGE) + 16u * (unsigned)(wq * 4 * 64 + lane);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // finishes group gi (k-slice 0 waves, after the barrier that made slice 1's partial visible): sum, affine, store
    auto finish = [&](int gi) {
        f32x4 p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = lds_read_b128(red0 + (unsigned)(q * 64 * 16));
        wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        float* cg = a.c + (long long)(g0 + gi) * a.c_gs + n0 + nb * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = fmaxf((acc[r] + p[r >> 2][r & 3]) * e_s + e_b, lo1);
            if (m < a.M && !(a.dbg & 8)) cg[(size_t)m * a.ldc] = v;
        }
    };

    int slot = 0, ku = 0, gi = 0;
    for (int st = 0; st < nst; ++st) {
        // Stage st has landed when only the loads of the NSR - 2 younger stages (issued, live or zero-sourced, in the iterations before this one)
        // are pending.  Stores of a finished group may still be in flight; vmcnt counts them too and they may complete out of order with the loads,
        // but pending stores only make this wait longer: loads return in order, so "at most (NSR - 2) LPS operations pending" implies that no load
        // older than those is among them.
        wait_vmcnt<(NSR - 2) * LPS>();
        wait_lgkmcnt<0>();                            // (slice 1's partial of the group that just ended is written)
        __builtin_amdgcn_s_barrier();                 // every wave's pieces of stage st are in LDS; every wave has read stage st - 1 (and scratch is published)
        {
            int fill = slot + NSR - 1;
            if (fill >= NSR) fill -= NSR;             // the slot read in iteration st - 1: free since the barrier
            issue_begin(st + NSR - 1, fill);          // (its LPS instructions follow one by one behind the MFMA groups below)
        }
        if (kq == 0 && ku == 0 && st > 0) {
            finish(gi - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        const unsigned Ab = a_base + (unsigned)(slot * STAGE * 4), Bb = b_base + (unsigned)(slot * STAGE * 4);
        f32x4 fa[2], fb[2];
        fa[0] = lds_read_b128(Ab + coff[0]);
        fb[0] = lds_read_b128(Bb + coff[0]);
#pragma unroll
        for (int sub = 0; sub < BK / 8; ++sub) {
            const int cur = sub & 1, nxt = cur ^ 1;
            if (sub + 1 < BK / 8) {
                fa[nxt] = lds_read_b128(Ab + coff[sub + 1]);
                fb[nxt] = lds_read_b128(Bb + coff[sub + 1]);
                wait_lgkmcnt<2>();
            } else {
                wait_lgkmcnt<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef VIDC_WGEMM_ATTRIB      // (attribution build only: the stream without its MFMAs)
            if (a.dbg & 1) { acc[0] += fa[cur].x + fb[cur].y + fa[cur].z + fb[cur].w; continue; }
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur].w, acc, 0, 0, 0);
#pragma unroll
            for (int j = sub; j < LPS; j += BK / 8) issue_one(j);      // behind MFMA group `sub`: instructions sub, sub + 4, ... of the look-ahead stage
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++ku == KS) {                             // the group's last stage: slice 1 publishes its partial (visible behind the next barrier) and starts over
            ku = 0;
            ++gi;
            if (kq == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 t;
                    t.x = acc[4 * q]; t.y = acc[4 * q + 1]; t.z = acc[4 * q + 2]; t.w = acc[4 * q + 3];
                    lds_write_b128(red0 + (unsigned)(q * 64 * 16), t);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
        }
        if (++slot == NSR) slot = 0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the zero-sourced look-ahead stages; slice 1's last partial)
    __builtin_amdgcn_s_barrier();
    if (kq == 0) finish(ng - 1);
}

template <int NB, int NSR>
int launch_cfg(const WArgs& a, int padded_chunks, hipStream_t st) {
    typedef Cfg<NB, NSR> C_;
    static bool attr_set[64] = {};
    int dev = 0;
    VIDC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgemm_stream_kernel<NB, NSR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C_::LDS_BYTES));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((wgemm_stream_kernel<NB, NSR>), dim3((unsigned)(padded_chunks * a.tiles_n)), dim3(64 * C_::NWV), C_::LDS_BYTES, st, a);
    VIDC_CHECK_LAUNCH("wgemm_stream_kernel");
    return VIDC_OK;
}

}  // namespace

namespace vidc {

// Launch behind vidc_conv2d_bn_act for the streamed tiles (csrc/conv_mfma.hip validates the common fields first).
// d->splitk: number of group chunks (0 / 1 = automatic: about two (NB = 1) / one (NB = 2) workgroups per CU).
int launch_wgemm_stream(const vidc_conv_desc& d, hipStream_t st) {
    const int NB = d.tile == VIDC_TILE_G96x64_STREAM3 ? 2 : 1, BN = 32 * NB;
    const long long M = (long long)d.B * d.Ho * d.Wo;
    VIDC_REQUIRE(d.precision == VIDC_PREC_FP32, VIDC_ERR_SHAPE, "conv (streamed tile): fp32 arithmetic only");
    VIDC_REQUIRE(d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0, VIDC_ERR_SHAPE, "conv (streamed tile): 1x1 / stride 1 / pad 0 only");
    VIDC_REQUIRE(M <= BM, VIDC_ERR_SHAPE, "conv (streamed tile): at most %d rows (got %lld)", BM, M);
    VIDC_REQUIRE(d.Cin % (BK * WK) == 0 && d.Cin >= 2 * BK * WK && d.Cout % BN == 0, VIDC_ERR_SHAPE,
                 "conv (streamed tile): Cin %% 64 == 0, Cin >= 128 and Cout %% %d == 0", BN);
    VIDC_REQUIRE(!(d.flags & ~VIDC_RELU1), VIDC_ERR_SHAPE, "conv (streamed tile): flags 0x%x not supported (one affine + ReLU only)", d.flags);
    VIDC_REQUIRE(d.p_gs == 0, VIDC_ERR_SHAPE, "conv (streamed tile): one scale / shift vector for all groups (p_gs == 0)");
    const long long a_bytes = M * d.ldx * 4, b_bytes = (long long)d.groups * d.w_gs * 4;
    VIDC_REQUIRE(a_bytes < (1ll << 31) && b_bytes < (1ll << 31) && (long long)d.groups * d.x_gs * 4 < (1ll << 31), VIDC_ERR_SHAPE,
                 "conv (streamed tile): operands must stay below 2 GiB (32-bit buffer offsets)");
    WArgs a;
    a.a = d.x; a.b = d.w; a.c = d.y; a.scale = d.scale1; a.shift = d.shift1;
    a.M = (int)M; a.N = d.Cout; a.K = d.Cin; a.lda = d.ldx; a.ldc = d.ldy; a.groups = d.groups; a.tiles_n = d.Cout / BN;
    a.relu = (d.flags & VIDC_RELU1) ? 1 : 0;
    a.a_gs = d.x_gs; a.b_gs = d.w_gs; a.c_gs = d.y_gs;
    a.a_bytes = (unsigned)a_bytes; a.b_bytes = (unsigned)b_bytes;
    static const int dbg = [] { const char* e = getenv("VIDC_WGEMM_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    const int target = NB == 2 ? 256 : 512;
    int chunks = d.splitk > 1 ? d.splitk : (target + a.tiles_n - 1) / a.tiles_n;
    if (chunks > d.groups) chunks = d.groups;
    if (chunks < 1) chunks = 1;
    a.chunks = chunks;
    const int padded = (chunks + 7) / 8 * 8;
    return NB == 2 ? launch_cfg<2, 3>(a, padded, st) : launch_cfg<1, 2>(a, padded, st);
}

}  // namespace vidc
