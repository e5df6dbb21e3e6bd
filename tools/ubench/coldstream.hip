// Micro-benchmark: HBM-cold weight streaming as the small conv layers do it.  160 workgroups; 10 consecutive workgroups share one
// "n-tile" (64 weight rows); every stage each workgroup DMAs the next 128 B x 2 k-slices of its 64 rows (16 KiB) into an LDS
// ring with DEPTH stages in flight.  Layout A: row-major rows of K*4 bytes (stage s touches 64 places 9216 B apart);
// layout B: tile-major (the 64 x 256 B of a stage are contiguous, consecutive stages follow each other).
//   hipcc --offload-arch=gfx950 -O3 -o coldstream coldstream.hip && ./coldstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;

template <int DEPTH>
__global__ void __launch_bounds__(256) k(const float* src, int layout, int row_bytes, int stages, long long tile_bytes, long long* out, float* sink, int sharers) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    // XCD-aware work-item mapping of the conv kernel: workgroup w (XCD w % 8) takes item v of a contiguous per-XCD range
    const int W = gridDim.x, w = blockIdx.x, xcd = w & 7, jj = w >> 3, q = W >> 3, r = W & 7;
    const int v = xcd * q + (xcd < r ? xcd : r) + jj;
    const int ntile = v / sharers;
    const unsigned base = (unsigned)(ntile * tile_bytes);
    // 16 wave-instructions per stage (64 rows x 2 slices x 128 B = 16 KiB), 4 per wave
    unsigned off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = j * 4 + wave;            // 0..15: slice = q / 8, row block = q % 8
        const int row = (q & 7) * 8 + (lane >> 3), slice = q >> 3;
        if (layout == 0) off[j] = base + row * row_bytes + slice * 128 + (lane & 7) * 16;          // row-major; stage advances 256 B along the row
        else if (layout == 1) off[j] = base + (slice * 64 + row) * 128 + (lane & 7) * 16;           // tile-major; stage advances 16 KiB
        else if (layout == 2) off[j] = base + row * row_bytes + slice * (row_bytes / 2) + (lane & 7) * 16;   // row-major, k-slices = halves of K (128 B per stage each)
        else                  off[j] = base + slice * (unsigned)(tile_bytes / 2) + row * 128 + (lane & 7) * 16;   // tile-major per slice: two 8 KiB-per-stage streams
    }
    const unsigned step = layout == 0 ? 256u : layout == 1 ? 16384u : layout == 2 ? 128u : 8192u;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* dst = smem + (((s % (DEPTH + 1)) * 16 + j * 4 + wave) * 256);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, (int)(off[j] + s * step), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (smem[threadIdx.x] == 123.456f) sink[1] = 1.f;
}

int main() {
    const size_t cap = 1024u << 20;
    float* src; long long* out; float* sink; unsigned char* junk;
    CK(hipMalloc(&src, cap)); CK(hipMemset(src, 0, cap));
    CK(hipMalloc(&junk, 512u << 20));
    CK(hipMalloc(&out, 8192 * 8)); CK(hipMalloc(&sink, 64));
    const int wgs = 160, stages = 36, row_bytes = 9216;
    const long long tile_bytes = 64ll * row_bytes;       // 576 KiB per n-tile
    auto run = [&](auto kern, int depth, int layout, int sharers, bool evict) -> int {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            if (evict) CK(hipMemset(junk, rep, 512u << 20));     // evict L2 / Infinity Cache
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), (depth + 1) * 16384, 0, src, layout, row_bytes, stages, tile_bytes, out, sink, sharers);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(wgs);
            CK(hipMemcpy(h.data(), out, wgs * sizeof(long long), hipMemcpyDeviceToHost));
            double cyc = 0; for (auto v : h) cyc += v; cyc /= wgs;
            if (cyc < best) best = cyc;
        }
        printf("%-10s depth %d sharers %3d %-4s: %6.0f clk per 16 KiB stage  (%.1f B/clk/CU, %.2f TB/s of unique bytes chip-wide)\n", (layout == 0 ? "row-major" : layout == 1 ? "tile-major" : layout == 2 ? "row/2halves" : "tile/2halves"), depth,
               sharers, evict ? "cold" : "hot", best / stages, 16384.0 * stages / best, (160.0 / sharers) * 16384 * stages / (best / 2.1e9) / 1e12);
        return 0;
    };
    for (int layout : {0, 1, 2, 3})
        for (int sharers : {10, 20})
            if (run(k<4>, 4, layout, sharers, true)) return 1;
    return 0;
}
