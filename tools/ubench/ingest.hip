// Micro-benchmark: how many bytes per clock can one CU pull from L2 (a) by LDS-DMA (buffer_load ... lds, 16 B/lane) and
// (b) by buffer_load_dwordx4 into VGPRs?  Every workgroup re-reads the same `span` bytes (L2-resident after the first pass).
//   hipcc --offload-arch=gfx950 -O3 -o ingest ingest.hip && ./ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MODE>   // MODE 0: LDS-DMA, 1: VGPR loads
__global__ void __launch_bounds__(512) ingest(const float* src, int span_bytes, int iters, long long* out, float* sink, int wg_stride_bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    const unsigned base = (unsigned)(((long long)blockIdx.x * wg_stride_bytes) % span_bytes);
    unsigned off = base + (wave * 64 + lane) * 16;
    const unsigned step = nw * 1024;
    f32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (MODE == 0) {
                float* dst = smem + (wave * 8 + (d & 7)) * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, (int)off, 0, 0, 0);
            } else {
                f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
                acc += v;
            }
            off += step;
            if (off >= (unsigned)span_bytes) off -= span_bytes;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (MODE == 1 && acc.x == 123.456f) sink[0] = acc.x + acc.y + acc.z + acc.w;
    if (MODE == 0 && smem[threadIdx.x] == 123.456f) sink[1] = 1.f;
}

template <int DEPTH, int MODE>
int run(const char* name, int threads, int wgs, int span, int iters, int lds_bytes, const float* src, long long* out, float* sink, int wg_stride) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ingest<DEPTH, MODE>), dim3(wgs), dim3(threads), lds_bytes, 0, src, span, iters, out, sink, wg_stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((ingest<DEPTH, MODE>), dim3(wgs), dim3(threads), lds_bytes, 0, src, span, iters, out, sink, wg_stride);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(wgs);
    CK(hipMemcpy(h.data(), out, wgs * sizeof(long long), hipMemcpyDeviceToHost));
    double cyc = 0; for (auto v : h) cyc += v; cyc /= wgs;
    const double bytes_wg = (double)iters * DEPTH * (threads / 64) * 1024.0;
    printf("%-8s thr %3d wgs %4d lds %3dKB span %5dKB stride %4dKB depth %2d: %.1f B/clk/WG  (%.0f cyc)  chip %.2f TB/s (event %.1f us)\n", name, threads, wgs,
           lds_bytes >> 10, span >> 10, wg_stride >> 10, DEPTH, bytes_wg / cyc, cyc, bytes_wg * wgs / (ms * 1e-3) / 1e12, ms * 1e3);
    return 0;
}

int main() {
    const int cap = 64 << 20;
    float* src; long long* out; float* sink;
    CK(hipMalloc(&src, cap)); CK(hipMemset(src, 0, cap));
    CK(hipMalloc(&out, 8192 * 8)); CK(hipMalloc(&sink, 64));
    const int iters = 200;
    for (int span : {256 << 10, 2 << 20}) {
        for (int threads : {64, 256, 512}) {
            // one WG per CU (LDS 100 KB forces 1/CU), then two per CU (LDS 70 KB)
            run<8, 0>("dma", threads, 256, span, iters, 100 << 10, src, out, sink, 0);
            run<16, 0>("dma", threads, 256, span, iters, 100 << 10, src, out, sink, 0);
            run<8, 0>("dma2/cu", threads, 512, span, iters, 70 << 10, src, out, sink, 0);
            run<8, 1>("vgpr", threads, 256, span, iters, 100 << 10, src, out, sink, 0);
            run<16, 1>("vgpr", threads, 256, span, iters, 100 << 10, src, out, sink, 0);
        }
    }
    // distinct data per WG (no sharing between WGs; 16 MB total -> per-XCD L2 holds its 2 MB share)
    run<8, 0>("dma-own", 256, 256, 16 << 20, iters, 100 << 10, src, out, sink, 64 << 10);
    run<16, 0>("dma-own", 256, 256, 16 << 20, iters, 100 << 10, src, out, sink, 64 << 10);
    run<16, 1>("vgpr-own", 256, 256, 16 << 20, iters, 100 << 10, src, out, sink, 64 << 10);
    // one WG alone on the chip: per-CU ceiling without any L2 contention
    run<16, 0>("dma-solo", 256, 1, 256 << 10, iters, 100 << 10, src, out, sink, 0);
    run<16, 0>("dma-solo", 512, 1, 256 << 10, iters, 100 << 10, src, out, sink, 0);
    run<16, 1>("vgpr-solo", 256, 1, 256 << 10, iters, 100 << 10, src, out, sink, 0);
    run<16, 0>("dma-8cu", 256, 8, 256 << 10, iters, 100 << 10, src, out, sink, 0);
    run<16, 0>("dma-64cu", 256, 64, 256 << 10, iters, 100 << 10, src, out, sink, 0);
    return 0;
}
