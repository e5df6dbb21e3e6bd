// Micro-benchmark: LDS-DMA ingest rate of one CU's worth of waves when the 8 rows of a wave-instruction (8 lanes x 16 B = one
// 128-byte row each) lie `stride` bytes apart -- the access pattern of the conv kernel's operand tiles (row = one output
// channel's K-run or one pixel's channel-run).  All workgroups read the same rows (L2-resident), like the m-tiles of a layer
// sharing a weight tile.    hipcc --offload-arch=gfx950 -O3 -o stride stride.hip && ./stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;

// each wave-instruction: rows r0..r0+7 (lane>>3), 128 B at column offset `col`; a "stage" = ROWS rows x 128 B; next stage: col += 128
template <int ROWS>
__global__ void __launch_bounds__(256) k(const float* src, int stride, int ncols, int iters, long long* out, float* sink, int wg_rows_off) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    constexpr int PER_WAVE = ROWS / 8 / 4;          // instructions per wave per stage (4 waves)
    const int row_base = (blockIdx.x * wg_rows_off) % 4096;
    unsigned off[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) off[j] = (unsigned)((row_base + (j * 4 + wave) * 8 + (lane >> 3)) * stride + (lane & 7) * 16);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int col = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            float* dst = smem + ((it & 3) * ROWS + (j * 4 + wave) * 8) * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, (int)(off[j] + col * 128), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
        if (++col == ncols) col = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (smem[threadIdx.x] == 123.456f) sink[1] = 1.f;
}

int main() {
    const size_t cap = 256u << 20;
    float* src; long long* out; float* sink;
    CK(hipMalloc(&src, cap)); CK(hipMemset(src, 0, cap));
    CK(hipMalloc(&out, 8192 * 8)); CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int iters = 400, wgs = 160;
    // stride in bytes between consecutive rows; ncols = 128-byte columns per row that are cycled through
    struct { int stride, ncols; const char* what; } cases[] = {
        {128, 1, "rows contiguous (tile-major packing), 1 column"},
        {9216, 72, "stride 9216 B  (K=2304 weights / activations ld=2304)"},
        {4096, 32, "stride 4096 B  (K=1024 weights; activations with 1024 channels)"},
        {1024, 8, "stride 1024 B  (K=256)"},
        {4096 + 128, 32, "stride 4224 B  (K=1024 padded by one unit)"},
        {9216 + 128, 72, "stride 9344 B  (K=2304 padded by one unit)"},
        {12288, 96, "stride 12288 B"},
    };
    for (auto& c : cases) {
        for (int share : {0, 96}) {       // 0: every workgroup reads the same 96 rows; 96: every workgroup its own rows
            hipLaunchKernelGGL((k<96>), dim3(wgs), dim3(256), 100 << 10, 0, src, c.stride, c.ncols, iters, out, sink, share);
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL((k<96>), dim3(wgs), dim3(256), 100 << 10, 0, src, c.stride, c.ncols, iters, out, sink, share);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(wgs);
            CK(hipMemcpy(h.data(), out, wgs * sizeof(long long), hipMemcpyDeviceToHost));
            double cyc = 0; for (auto v : h) cyc += v; cyc /= wgs;
            printf("%-62s %-10s %6.1f B/clk/CU  (%.0f clk per 12 KiB stage)\n", c.what, share ? "own rows" : "shared", 96.0 * 128 * iters / cyc, cyc / iters);
        }
    }
    return 0;
}
