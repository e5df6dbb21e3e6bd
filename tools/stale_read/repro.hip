// Stand-alone reproduction attempt of the round-5 stale-read hazard (VERDICT r5 "weak 1"), outside the package and outside torch:
// L lanes (HIP streams), each with its own input buffers and two captured graphs per tick.  Per tick and lane, as pipeline._GroupLane does:
//   put:   D2D copies rewrite the lane's inputs (image slots, gravity words, a few more tensors);
//   seg 0: graph [params kernel: input words -> 32-word record] [consumer: reads the record wave-uniformly and the image with the
//          fused stem's patch pattern (or the warp kernel's banded pattern)] [NF0 streaming filler kernels];
//   then a small kernel + 1.3 KB device->host read the host WAITS for (the candidate counts), seg 1 (NF1 fillers) launched before the wait;
//   next put of the lane only after that wait.
// Every word the consumer reads is written out with a tag (item << 20 | index), so a stale word names the item it came from.
//
// Build:  hipcc --offload-arch=gfx950 -O2 -o tools/stale_read/repro tools/stale_read/repro.hip
// Run:    tools/stale_read/repro --lanes 3 --F 4 --items 2000 --exec graph --img plain --rec scalar   (see main() for the rest)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int H = 240, W = 320, C = 3, PLANE = H * W, IMG = C * PLANE;      // 230 400 words per image
constexpr int REC = 32;
constexpr int Ho = H / 2, Wo = W / 2, PIX = 64, PW = 2 * PIX + 1, NBX = (Wo + PIX - 1) / PIX;    // stem: 3x3 stride 2 pad 1
constexpr int READS_WG = C * 3 * PW, NWG = Ho * NBX, READS = NWG * READS_WG;                          // words read (and logged) per image

__global__ void gen_pool(uint32_t* pool, int n_items) {
    const size_t i = blockIdx.y, w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < (size_t)IMG) pool[i * IMG + w] = ((uint32_t)i << 20) | (uint32_t)w;
}
__global__ void gen_words(uint32_t* g, int n_items) {      // 4 "gravity" words per item
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_items) for (int j = 0; j < 4; ++j) g[i * 4 + j] = ((uint32_t)i << 20) | ((uint32_t)j << 8);
}

// producer node: one thread per slot, like warp_params_kernel (one workgroup of 64)
__global__ void params_kernel(const uint32_t* __restrict__ g, int F, uint32_t* __restrict__ rec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= F) return;
    for (int k = 0; k < REC; ++k) rec[b * REC + k] = g[b * 4 + (k & 3)] + (uint32_t)k;
}

enum { IMG_PLAIN = 0, IMG_SYS = 1, IMG_INV = 2, IMG_NT = 3 };
enum { REC_SCALAR = 0, REC_VECTOR = 1, REC_SYS = 2, REC_SCALAR_DCINV = 3 };

template <int IMODE>
__device__ __forceinline__ uint32_t ld_img(const uint32_t* p) {
    if constexpr (IMODE == IMG_SYS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if constexpr (IMODE == IMG_NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// consumer, pattern 0: the fused stem's patch loader (every input row is read by two workgroups, every workgroup reads 3 x 129 x C words)
template <int IMODE, int RMODE>
__global__ void __launch_bounds__(256)
consumer_stem(const uint32_t* __restrict__ x, const uint32_t* __restrict__ rec, uint32_t* __restrict__ out_reads, uint32_t* __restrict__ out_rec) {
    if constexpr (IMODE == IMG_INV) asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (RMODE == REC_SCALAR_DCINV) { __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    const int b = blockIdx.z, oy = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    const int wg = oy * NBX + bx;
    const uint32_t* xb = x + (size_t)b * IMG;
    const int ix_base = bx * PIX * 2 - 1, iy_base = oy * 2 - 1;
    uint32_t* o = out_reads + ((size_t)b * NWG + wg) * READS_WG;
    for (int e = tid; e < READS_WG; e += 256) {
        const int c = e / (3 * PW), r = (e / PW) % 3, i = e % PW;
        const int iy = iy_base + r, ix = ix_base + i;
        uint32_t v = 0xFFFFFFFFu;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = ld_img<IMODE>(xb + c * PLANE + iy * W + ix);
        o[e] = v;
    }
    uint32_t* orr = out_rec + ((size_t)b * NWG + wg) * REC;
    if constexpr (RMODE == REC_SCALAR || RMODE == REC_SCALAR_DCINV) {
        const uint32_t* r = rec + (size_t)b * REC;          // wave-uniform address: s_load
        uint32_t acc[REC];
#pragma unroll
        for (int k = 0; k < REC; ++k) acc[k] = r[k];
        if (tid < REC) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < REC; ++k) v = (tid == k) ? acc[k] : v;
            orr[tid] = v;
        }
    } else if constexpr (RMODE == REC_VECTOR) {
        if (tid < REC) orr[tid] = __builtin_nontemporal_load(rec + (size_t)b * REC + tid);
    } else {
        if (tid < REC) orr[tid] = __hip_atomic_load(rec + (size_t)b * REC + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// consumer, pattern 2: pattern 0 staged through LDS as the stem does (loads -> ds_write -> barrier -> ds_read -> store)
template <int IMODE, int RMODE>
__global__ void __launch_bounds__(256)
consumer_stem_lds(const uint32_t* __restrict__ x, const uint32_t* __restrict__ rec, uint32_t* __restrict__ out_reads, uint32_t* __restrict__ out_rec) {
    __shared__ uint32_t patch[READS_WG + 3];
    const int b = blockIdx.z, oy = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    const int wg = oy * NBX + bx;
    const uint32_t* xb = x + (size_t)b * IMG;
    const int ix_base = bx * PIX * 2 - 1, iy_base = oy * 2 - 1;
    uint32_t* o = out_reads + ((size_t)b * NWG + wg) * READS_WG;
    for (int e = tid; e < 3 * PW; e += 256) {
        const int r = e / PW, i = e - r * PW;
        const int iy = iy_base + r, ix = ix_base + i;
        const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
#pragma unroll
        for (int c = 0; c < C; ++c) patch[(c * 3 + r) * PW + i] = in ? ld_img<IMODE>(xb + c * PLANE + iy * W + ix) : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int e = tid; e < READS_WG; e += 256) o[e] = patch[e];
    if (tid < REC) out_rec[((size_t)b * NWG + wg) * REC + tid] = __hip_atomic_load(rec + (size_t)b * REC + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// consumer, pattern 1: warp_fwd_kernel's shape -- one thread per pixel, XCD-banded block order, C planes, the record read wave-uniformly
template <int IMODE, int RMODE>
__global__ void __launch_bounds__(256)
consumer_band(const uint32_t* __restrict__ x, const uint32_t* __restrict__ rec, uint32_t* __restrict__ out_reads, uint32_t* __restrict__ out_rec) {
    if constexpr (IMODE == IMG_INV) asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (RMODE == REC_SCALAR_DCINV) { __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, per = (gx >> 3) * gridDim.y;
    const unsigned idx = (L & 7u) * per + (L >> 3);
    const unsigned by = idx / gx, bxx = idx - by * gx;
    const int b = (int)by, pix = (int)(bxx * 256 + threadIdx.x);
    uint32_t rsum = 0;
    if constexpr (RMODE == REC_SCALAR || RMODE == REC_SCALAR_DCINV) {
        const uint32_t* r = rec + (size_t)b * REC;
        rsum = r[0];                                          // all 32 words carry the item in the top 12 bits; one is enough here
    } else if constexpr (RMODE == REC_VECTOR) rsum = __builtin_nontemporal_load(rec + (size_t)b * REC);
    else rsum = __hip_atomic_load(rec + (size_t)b * REC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) out_rec[((size_t)b * NWG) * REC + (bxx % (NWG * REC))] = rsum;
    if (pix >= PLANE) return;
    for (int c = 0; c < C; ++c) out_reads[(size_t)b * READS + c * PLANE + pix] = ld_img<IMODE>(x + (size_t)b * IMG + c * PLANE + pix);
}

// consumer, pattern 3 ("war"): the instruction pattern hipcc generated in the failing stem -- a global_load whose 64-bit ADDRESS register pair is
// overwritten by the very next VALU instruction(s) (stem_conv_kernel<3,true,3>: `global_load_dword v39, v[38:39], off; s_waitcnt vmcnt(6);
// v_mul_f32 v38, ...`).  If the load of some lanes picks up the NEW register contents it reads from `poison` (tag 0xDEADxxxx) instead of the image.
// GAP = independent VALU instructions between the load and the overwrite.
template <int GAP>
__global__ void __launch_bounds__(256)
consumer_war(const uint32_t* __restrict__ x, const uint32_t* __restrict__ poison, uint32_t* __restrict__ out_reads) {
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, per = (gx >> 3) * gridDim.y;
    const unsigned idx = (L & 7u) * per + (L >> 3);
    const unsigned by = idx / gx, bxx = idx - by * gx;
    const int b = (int)by, pix = (int)(bxx * 256 + threadIdx.x);
    if (pix >= PLANE) return;
    const uint32_t* pz = poison + (pix & 4095);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t* src = x + (size_t)b * IMG + c * PLANE + pix;
        uint32_t v;
        unsigned long long a = (unsigned long long)src;
        const unsigned long long z = (unsigned long long)pz;
        float f0 = (float)pix, f1 = 1.5f;
        if constexpr (GAP == 0)
            asm volatile("global_load_dword %0, %1, off\n\tv_mov_b64 %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "+v"(a) : "v"(z) : "memory");
        else if constexpr (GAP == 1)
            asm volatile("global_load_dword %0, %1, off\n\tv_mul_f32 %2, %2, %4\n\tv_mov_b64 %1, %3\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "+v"(a), "+v"(f0) : "v"(z), "v"(f1) : "memory");
        else if constexpr (GAP == 2)
            asm volatile("global_load_dword %0, %1, off\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mov_b64 %1, %3\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "+v"(a), "+v"(f0) : "v"(z), "v"(f1) : "memory");
        else
            asm volatile("global_load_dword %0, %1, off\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mov_b64 %1, %3\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "+v"(a), "+v"(f0) : "v"(z), "v"(f1) : "memory");
        out_reads[(size_t)b * READS + c * PLANE + pix] = v + (a == 1ull ? 1u : 0u) + (f0 == 0.123f ? 1u : 0u);
    }
}

// consumer, pattern 4 ("raw"): four loads in flight, counted waits, every result consumed by the VALU instruction right behind its wait -- the
// shape of the failing stem's tap loads (`s_waitcnt vmcnt(6); v_mul_f32 v40, v34, v54`).  The destination registers hold a poison word before the
// loads are issued: a lane whose data had not landed when the wait let the wave through reports 0xBAD0xxxx.
// MODE 0: counted waits vmcnt(3..0); MODE 1: one vmcnt(0) for all four, then the moves (the form that never failed).
template <int MODE>
__global__ void __launch_bounds__(256)
consumer_raw(const uint32_t* __restrict__ x, uint32_t* __restrict__ out_reads) {
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, per = (gx >> 3) * gridDim.y;
    const unsigned idx = (L & 7u) * per + (L >> 3);
    const unsigned by = idx / gx, bxx = idx - by * gx;
    const int b = (int)by, pix = (int)(bxx * 256 + threadIdx.x);
    if (pix >= PLANE) return;
    const uint32_t* p0 = x + (size_t)b * IMG + pix;
    const uint32_t *p1 = p0 + PLANE, *p2 = p0 + 2 * PLANE;
    uint32_t r0 = 0xBAD00000u | (threadIdx.x & 63), r1 = r0 | 0x100u, r2 = r0 | 0x200u, r3 = r0 | 0x300u, o0, o1, o2, o3;
    if constexpr (MODE == 0)
        asm volatile("global_load_dword %4, %8, off\n\tglobal_load_dword %5, %9, off\n\tglobal_load_dword %6, %10, off\n\tglobal_load_dword %7, %8, off\n\t"
                     "s_waitcnt vmcnt(3)\n\tv_mov_b32 %0, %4\n\ts_waitcnt vmcnt(2)\n\tv_mov_b32 %1, %5\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %2, %6\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 %3, %7"
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(p0), "v"(p1), "v"(p2) : "memory");
    else
        asm volatile("global_load_dword %4, %8, off\n\tglobal_load_dword %5, %9, off\n\tglobal_load_dword %6, %10, off\n\tglobal_load_dword %7, %8, off\n\t"
                     "s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(p0), "v"(p1), "v"(p2) : "memory");
    uint32_t* o = out_reads + (size_t)b * READS + pix;
    o[0] = o0; o[PLANE] = o1; o[2 * PLANE] = o2; o[3 * PLANE] = o3;
}

__global__ void __launch_bounds__(256) filler(float* __restrict__ buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = buf[i] * 1.0001f + 1.0f;
}
__global__ void small_counts(const uint32_t* __restrict__ a, uint32_t* __restrict__ counts) {
    counts[threadIdx.x] = a[threadIdx.x] + 1u;
}
__global__ void copy_kernel(const uint32_t* __restrict__ s, uint32_t* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

// ---- noise: a co-resident kernel of another stream (round 6: the package-level bisect named the bf16x3 conv kernel as the neighbour
// that makes the plain-load consumer read wrong quads) -----------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// MODE bit 0: bf16 MFMAs (v_mfma_f32_32x32x16_bf16), bit 1: fp32 MFMAs (v_mfma_f32_32x32x2_f32), bit 2: ds_read_b128 fragments from LDS in the
// loop, bit 3: 16-byte-per-lane LDS-DMA refills (buffer_load ... lds) from a global buffer in the loop
template <int MODE>
__global__ void __launch_bounds__(256) noise_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float nsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4v fa = {1.f, 2.f, 3.f, 4.f}, fb = {0.5f, 0.25f, 0.125f, 1.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 24, 0x00020000);
    for (int i = tid; i < 8192; i += 256) nsm[i] = (float)i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE & 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(nsm + 8192 + wave * 256), 16, (int)((lane * 16 + (it & 63) * 1024) ), 0, 0, 0);
        }
        if constexpr (MODE & 4) {
            fa = *reinterpret_cast<const f32x4v*>(nsm + ((lane * 4 + it * 64) & 8188));
            fb = *reinterpret_cast<const f32x4v*>(nsm + ((lane * 4 + it * 64 + 2048) & 8188));
        }
        if constexpr (MODE & 1) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, fa), b = __builtin_bit_cast(bf16x8, fb);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc1, 0, 0, 0);
        }
        if constexpr (MODE & 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc1, 0, 0, 0);
        }
        if constexpr (!(MODE & 3)) { fa.x = fa.x * 1.0001f + fb.y; }
    }
    if constexpr (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = fa.x;
    for (int k = 0; k < 16; ++k) r += acc0[k] + acc1[k];
    if (r == 123.456f) sink[tid] = r;
}

struct Opt {
    int lanes = 3, F = 4, items = 2000, nf0 = 60, nf1 = 30, pattern = 0, hostsync = 0, copykernel = 0, graph = 1, img = IMG_PLAIN, rec = REC_SCALAR, verbose = 0;
    int noise = -1, noise_iters = 4000, noise_wgs = 1024, noise_lds_kb = 48, gap = 0;
    size_t fill_words = 8u << 20;
};

struct Lane {
    hipStream_t st;
    uint32_t *x, *x2, *g, *a, *ds, *homo, *rec, *out_reads, *out_rec, *counts, *counts_host, *up, *up_host;
    float* fill;
    hipGraphExec_t seg0 = nullptr, seg1 = nullptr;
    hipEvent_t ev;
    std::vector<int> cur;      // items of the group whose seg 0 was launched last
    bool have_prev = false;
};

template <int IM, int RM>
static void launch_consumer(const Opt& o, Lane& l) {
    if (o.pattern == 0) hipLaunchKernelGGL((consumer_stem<IM, RM>), dim3(NBX, Ho, o.F), dim3(256), 0, l.st, l.x, l.rec, l.out_reads, l.out_rec);
    else if (o.pattern == 2) hipLaunchKernelGGL((consumer_stem_lds<IM, RM>), dim3(NBX, Ho, o.F), dim3(256), 0, l.st, l.x, l.rec, l.out_reads, l.out_rec);
    else hipLaunchKernelGGL((consumer_band<IM, RM>), dim3(((PLANE + 255) / 256 + 7) / 8 * 8, o.F), dim3(256), 0, l.st, l.x, l.rec, l.out_reads, l.out_rec);
}
template <int IM>
static void launch_consumer_r(const Opt& o, Lane& l) {
    switch (o.rec) {
        case REC_SCALAR: launch_consumer<IM, REC_SCALAR>(o, l); break;
        case REC_VECTOR: launch_consumer<IM, REC_VECTOR>(o, l); break;
        case REC_SYS: launch_consumer<IM, REC_SYS>(o, l); break;
        default: launch_consumer<IM, REC_SCALAR_DCINV>(o, l); break;
    }
}
static uint32_t* g_poison = nullptr;
static void seg0_ops(const Opt& o, Lane& l) {
    hipLaunchKernelGGL(params_kernel, dim3(1), dim3(64), 0, l.st, l.g, o.F, l.rec);
    if (o.pattern == 3) {
        const dim3 g(((PLANE + 255) / 256 + 7) / 8 * 8, o.F), b(256);
        if (o.gap == 0) hipLaunchKernelGGL((consumer_war<0>), g, b, 0, l.st, l.x, g_poison, l.out_reads);
        else if (o.gap == 1) hipLaunchKernelGGL((consumer_war<1>), g, b, 0, l.st, l.x, g_poison, l.out_reads);
        else if (o.gap == 2) hipLaunchKernelGGL((consumer_war<2>), g, b, 0, l.st, l.x, g_poison, l.out_reads);
        else hipLaunchKernelGGL((consumer_war<4>), g, b, 0, l.st, l.x, g_poison, l.out_reads);
        for (int k = 0; k < o.nf0; ++k) hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, l.st, l.fill, o.fill_words);
        return;
    }
    if (o.pattern == 4) {
        const dim3 g(((PLANE + 255) / 256 + 7) / 8 * 8, o.F), b(256);
        if (o.gap == 0) hipLaunchKernelGGL((consumer_raw<0>), g, b, 0, l.st, l.x, l.out_reads);
        else hipLaunchKernelGGL((consumer_raw<1>), g, b, 0, l.st, l.x, l.out_reads);
        for (int k = 0; k < o.nf0; ++k) hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, l.st, l.fill, o.fill_words);
        return;
    }
    switch (o.img) {
        case IMG_PLAIN: launch_consumer_r<IMG_PLAIN>(o, l); break;
        case IMG_SYS: launch_consumer_r<IMG_SYS>(o, l); break;
        case IMG_INV: launch_consumer_r<IMG_INV>(o, l); break;
        default: launch_consumer_r<IMG_NT>(o, l); break;
    }
    for (int k = 0; k < o.nf0; ++k) hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, l.st, l.fill, o.fill_words);
}
static void seg1_ops(const Opt& o, Lane& l) {
    for (int k = 0; k < o.nf1; ++k) hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, l.st, l.fill, o.fill_words);
}
static hipGraphExec_t capture(const Opt& o, Lane& l, void (*ops)(const Opt&, Lane&)) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(l.st, hipStreamCaptureModeThreadLocal));
    ops(o, l);
    CK(hipStreamEndCapture(l.st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ge;
}
static void d2d(const Opt& o, Lane& l, void* dst, const void* src, size_t bytes) {
    if (o.copykernel) hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, l.st, (const uint32_t*)src, (uint32_t*)dst, bytes / 4);
    else CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, l.st));
}

int main(int argc, char** argv) {
    Opt o;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return std::string(argv[++i]); };
        if (k == "--lanes") o.lanes = atoi(val().c_str());
        else if (k == "--F") o.F = atoi(val().c_str());
        else if (k == "--items") o.items = atoi(val().c_str());
        else if (k == "--nf0") o.nf0 = atoi(val().c_str());
        else if (k == "--nf1") o.nf1 = atoi(val().c_str());
        else if (k == "--fill-mb") o.fill_words = (size_t)atoi(val().c_str()) << 18;
        else if (k == "--pattern") { auto v = val(); o.pattern = (v == "band") ? 1 : (v == "stemlds") ? 2 : (v == "war") ? 3 : (v == "raw") ? 4 : 0; }
        else if (k == "--hostsync") o.hostsync = atoi(val().c_str());
        else if (k == "--copy") { auto v = val(); o.copykernel = (v == "kernel"); }
        else if (k == "--exec") { auto v = val(); o.graph = (v == "graph"); }
        else if (k == "--img") { auto v = val(); o.img = v == "plain" ? IMG_PLAIN : v == "sys" ? IMG_SYS : v == "inv" ? IMG_INV : IMG_NT; }
        else if (k == "--rec") { auto v = val(); o.rec = v == "scalar" ? REC_SCALAR : v == "vector" ? REC_VECTOR : v == "sys" ? REC_SYS : REC_SCALAR_DCINV; }
        else if (k == "--verbose") o.verbose = 1;
        else if (k == "--noise") o.noise = atoi(val().c_str());
        else if (k == "--gap") o.gap = atoi(val().c_str());
        else if (k == "--noise-iters") o.noise_iters = atoi(val().c_str());
        else if (k == "--noise-wgs") o.noise_wgs = atoi(val().c_str());
        else if (k == "--noise-lds-kb") o.noise_lds_kb = atoi(val().c_str());
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (o.items > 4095) o.items = 4095;      // 12 tag bits
    const int F = o.F, L = o.lanes, N = o.items;
    uint32_t *pool, *gpool, *res_reads, *res_rec;
    CK(hipMalloc(&pool, (size_t)N * IMG * 4));
    CK(hipMalloc(&gpool, (size_t)N * 16));
    CK(hipMalloc(&res_reads, (size_t)N * READS * 4));
    CK(hipMalloc(&res_rec, (size_t)N * NWG * REC * 4));
    CK(hipMemset(res_reads, 0xEE, (size_t)N * READS * 4));
    CK(hipMemset(res_rec, 0xEE, (size_t)N * NWG * REC * 4));
    hipLaunchKernelGGL(gen_pool, dim3((IMG + 255) / 256, N), dim3(256), 0, 0, pool, N);
    hipLaunchKernelGGL(gen_words, dim3((N + 255) / 256), dim3(256), 0, 0, gpool, N);
    CK(hipDeviceSynchronize());

    {
        std::vector<uint32_t> pz(8192);
        for (int k = 0; k < 8192; ++k) pz[k] = 0xDEAD0000u | (uint32_t)k;
        CK(hipMalloc(&g_poison, 8192 * 4));
        CK(hipMemcpy(g_poison, pz.data(), 8192 * 4, hipMemcpyHostToDevice));
    }
    std::vector<Lane> lanes(L);
    // like reserve_lane_streams(): the lanes' streams first, one launch on each
    for (auto& l : lanes) CK(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
    for (auto& l : lanes) {
        CK(hipMalloc(&l.x, (size_t)F * IMG * 4)); CK(hipMalloc(&l.x2, (size_t)F * IMG * 4)); CK(hipMalloc(&l.g, F * 16)); CK(hipMalloc(&l.a, F * 16));
        CK(hipMalloc(&l.ds, (size_t)F * PLANE * 4)); CK(hipMalloc(&l.homo, (size_t)F * IMG * 4)); CK(hipMalloc(&l.rec, F * REC * 4));
        CK(hipMalloc(&l.out_reads, (size_t)F * READS * 4)); CK(hipMalloc(&l.out_rec, (size_t)F * NWG * REC * 4));
        CK(hipMalloc(&l.counts, 4096)); CK(hipHostMalloc(&l.counts_host, 4096)); CK(hipMalloc(&l.up, 4096)); CK(hipHostMalloc(&l.up_host, 4096));
        CK(hipMalloc(&l.fill, o.fill_words * 4));
        CK(hipMemset(l.x, 0, (size_t)F * IMG * 4)); CK(hipMemset(l.g, 0, F * 16)); CK(hipMemset(l.fill, 0, o.fill_words * 4)); CK(hipMemset(l.counts, 0, 4096));
        CK(hipEventCreateWithFlags(&l.ev, hipEventDisableTiming));
    }
    CK(hipDeviceSynchronize());
    for (auto& l : lanes) {
        seg0_ops(o, l); seg1_ops(o, l);      // warm-up outside capture
        CK(hipStreamSynchronize(l.st));
        if (o.graph) {
            l.seg0 = capture(o, l, seg0_ops); l.seg1 = capture(o, l, seg1_ops);
            CK(hipGraphLaunch(l.seg0, l.st)); CK(hipGraphLaunch(l.seg1, l.st));
            CK(hipStreamSynchronize(l.st));
        }
    }
    hipStream_t nst;
    CK(hipStreamCreateWithFlags(&nst, hipStreamNonBlocking));
    float *nsrc, *nsink;
    CK(hipMalloc(&nsrc, 1 << 24)); CK(hipMalloc(&nsink, 4096)); CK(hipMemset(nsrc, 0, 1 << 24));
    auto noise_launch = [&]() {
        if (o.noise < 0) return;
        const size_t lds = (size_t)o.noise_lds_kb * 1024;
        const dim3 g(o.noise_wgs), b(256);
#define NZ(M_) case M_: { static bool set = false; if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&noise_kernel<M_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; } \
                          hipLaunchKernelGGL((noise_kernel<M_>), g, b, lds, nst, nsrc, nsink, o.noise_iters); break; }
        switch (o.noise) { NZ(0) NZ(1) NZ(2) NZ(4) NZ(5) NZ(6) NZ(8) NZ(9) NZ(10) NZ(12) NZ(13) NZ(14) default: fprintf(stderr, "noise mode %d not instantiated\n", o.noise); exit(2); }
#undef NZ
    };
    hipEvent_t nev[2];
    CK(hipEventCreateWithFlags(&nev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&nev[1], hipEventDisableTiming));
    noise_launch(); CK(hipEventRecord(nev[0], nst)); noise_launch(); CK(hipEventRecord(nev[1], nst));
    int nturn = 0;
    auto noise_keep_going = [&]() {      // two noise launches in flight at any time
        if (o.noise < 0) return;
        if (hipEventQuery(nev[nturn]) == hipSuccess) { noise_launch(); CK(hipEventRecord(nev[nturn], nst)); nturn ^= 1; }
    };
    int taken = 0;
    auto start = [&](Lane& l) -> bool {      // put up to F items + seg 0
        std::vector<int> idx;
        if (taken >= N) return false;
        if (l.have_prev) d2d(o, l, l.x2, l.x, (size_t)F * IMG * 4);                      // dc_image.copy_(sn_image)
        for (int j = 0; j < F && taken < N; ++j, ++taken) {
            const int it = taken;
            d2d(o, l, l.x + (size_t)j * IMG, pool + (size_t)it * IMG, (size_t)IMG * 4);   // sn_image[slot].copy_(rgb)
            CK(hipMemcpyAsync(l.g + j * 4, gpool + (size_t)it * 4, 16, hipMemcpyDeviceToDevice, l.st));
            CK(hipMemcpyAsync(l.a + j * 4, gpool + (size_t)it * 4, 16, hipMemcpyDeviceToDevice, l.st));
            d2d(o, l, l.ds + (size_t)j * PLANE, pool + (size_t)it * IMG, (size_t)PLANE * 4);
            d2d(o, l, l.homo + (size_t)j * IMG, pool + (size_t)((it + 1) % N) * IMG, (size_t)IMG * 4);
            idx.push_back(it);
        }
        if (o.hostsync) CK(hipDeviceSynchronize());
        if (o.graph) CK(hipGraphLaunch(l.seg0, l.st)); else seg0_ops(o, l);
        // results of this group: stream-ordered copies out of the lane's output buffers
        for (size_t j = 0; j < idx.size(); ++j) {
            CK(hipMemcpyAsync(res_reads + (size_t)idx[j] * READS, l.out_reads + j * (size_t)READS, (size_t)READS * 4, hipMemcpyDeviceToDevice, l.st));
            CK(hipMemcpyAsync(res_rec + (size_t)idx[j] * NWG * REC, l.out_rec + j * (size_t)NWG * REC, (size_t)NWG * REC * 4, hipMemcpyDeviceToDevice, l.st));
        }
        l.cur = idx;
        return true;
    };
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventRecord(t0, 0));
    int live = 0;
    for (int k = 0; k < L; ++k) if (start(lanes[k])) ++live;
    std::vector<bool> active(L, false);
    for (int k = 0; k < live; ++k) active[k] = true;
    for (int p = 0;; ++p) {
        Lane& l = lanes[p % L];
        if (!active[p % L]) break;
        // "plane kernels" + the read the host waits for, seg 1 launched before the wait
        hipLaunchKernelGGL(small_counts, dim3(1), dim3(256), 0, l.st, l.rec, l.counts);
        CK(hipMemcpyAsync(l.counts_host, l.counts, 1300, hipMemcpyDeviceToHost, l.st));
        CK(hipEventRecord(l.ev, l.st));
        if (l.have_prev) { if (o.graph) CK(hipGraphLaunch(l.seg1, l.st)); else seg1_ops(o, l); }
        noise_keep_going();
        CK(hipEventSynchronize(l.ev));
        noise_keep_going();
        CK(hipMemcpyAsync(l.up, l.up_host, 800, hipMemcpyHostToDevice, l.st));            // the enrichment draws
        hipLaunchKernelGGL(small_counts, dim3(1), dim3(256), 0, l.st, l.up, l.counts);
        l.have_prev = true;
        if (!start(l)) { active[p % L] = false; if (o.graph) CK(hipGraphLaunch(l.seg1, l.st)); else seg1_ops(o, l); }
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1));
    float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));

    // ---- verification: every logged word must carry its own item's tag ----------------------------------------------------------------
    std::vector<uint32_t> hr((size_t)READS), hrec((size_t)NWG * REC);
    long bad_items = 0, bad_words = 0, bad_rec_items = 0, bad_rec_words = 0;
    std::map<int, long> delta_img, delta_rec;      // (expected item - seen item) -> words
    for (int it = 0; it < N; ++it) {
        CK(hipMemcpy(hr.data(), res_reads + (size_t)it * READS, (size_t)READS * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hrec.data(), res_rec + (size_t)it * NWG * REC, (size_t)NWG * REC * 4, hipMemcpyDeviceToHost));
        long bw = 0, br = 0;
        if (o.pattern == 0 || o.pattern == 2) {
            for (int wg = 0; wg < NWG; ++wg) {
                const int oy = wg / NBX, bx = wg % NBX;
                for (int e = 0; e < READS_WG; ++e) {
                    const int c = e / (3 * PW), r = (e / PW) % 3, i = e % PW, iy = oy * 2 - 1 + r, ix = bx * PIX * 2 - 1 + i;
                    uint32_t want = 0xFFFFFFFFu;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) want = ((uint32_t)it << 20) | (uint32_t)(c * PLANE + iy * W + ix);
                    const uint32_t got = hr[(size_t)wg * READS_WG + e];
                    if (got != want) { ++bw; delta_img[(got == 0xEEEEEEEEu) ? -99999 : it - (int)(got >> 20)]++; if (o.verbose && bw < 4) printf("  item %d wg %d e %d: want %08x got %08x\n", it, wg, e, want, got); }
                }
                for (int k = 0; k < REC; ++k) {
                    const uint32_t want = (((uint32_t)it << 20) | ((uint32_t)(k & 3) << 8)) + (uint32_t)k, got = hrec[(size_t)wg * REC + k];
                    if (got != want) { ++br; delta_rec[(got == 0xEEEEEEEEu) ? -99999 : it - (int)(got >> 20)]++; }
                }
            }
        } else {
            for (int w = 0; w < (o.pattern == 4 ? 4 * PLANE : IMG); ++w) {
                const uint32_t want = ((uint32_t)it << 20) | (uint32_t)(w >= IMG ? w - IMG : w), got = hr[w];
                if (got != want) { ++bw; delta_img[(got == 0xEEEEEEEEu) ? -99999 : (got >> 16) == 0xDEADu ? -77777 : (got >> 20) == 0xBADu ? -66666 : it - (int)(got >> 20)]++; if (o.verbose && bw < 6) printf("  item %d word %d (pix %d, lane %d): want %08x got %08x\n", it, w, w % PLANE, (w % PLANE) & 63, want, got); }
            }
            const int nb = o.pattern >= 3 ? 0 : ((PLANE + 255) / 256 + 7) / 8 * 8;
            for (int k = 0; k < nb && k < NWG * REC; ++k) {
                const uint32_t got = hrec[k];
                if ((got >> 20) != (uint32_t)it) { ++br; delta_rec[(got == 0xEEEEEEEEu) ? -99999 : it - (int)(got >> 20)]++; }
            }
        }
        if (bw) { ++bad_items; bad_words += bw; if (o.verbose) printf("item %d (group %d, lane %d): %ld stale image words\n", it, it / F, (it / F) % L, bw); }
        if (br) { ++bad_rec_items; bad_rec_words += br; if (o.verbose) printf("item %d (group %d, lane %d): %ld stale record words\n", it, it / F, (it / F) % L, br); }
    }
    printf("RESULT noise=%d lanes=%d F=%d items=%d exec=%s pattern=%s img=%d rec=%d copy=%s hostsync=%d nf0=%d nf1=%d : %.1f ms, %.0f items/s | image: %ld items / %ld words stale | record: %ld items / %ld words stale\n",
           o.noise, L, F, N, o.graph ? "graph" : "eager", o.pattern == 1 ? "band" : o.pattern == 2 ? "stemlds" : o.pattern == 3 ? "war" : o.pattern == 4 ? "raw" : "stem", o.img, o.rec, o.copykernel ? "kernel" : "memcpy", o.hostsync, o.nf0, o.nf1, ms, N / (ms * 1e-3),
           bad_items, bad_words, bad_rec_items, bad_rec_words);
    for (auto& kv : delta_img) printf("   image words from item (own - %d): %ld\n", kv.first, kv.second);
    for (auto& kv : delta_rec) printf("   record words from item (own - %d): %ld\n", kv.first, kv.second);
    return 0;
}
