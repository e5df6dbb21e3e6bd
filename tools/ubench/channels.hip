// Micro-benchmark: how does the rate of ONE HBM-cold stream depend on how its addresses are spread?  One workgroup per CU on
// NWG CUs, all reading the same data (one "stream", like the m-tiles of a small conv layer sharing a weight tile).  Each stage =
// 16 wave-loads of 1 KiB; load q of stage s reads from region (q % R) at offset s*chunk (+ position within the chunk), regions D
// bytes apart.  R = 1: a plain sequential stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;

__global__ void __launch_bounds__(256) k(const float* src, int R, long long D, int stages, long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    const int per_region = 16 / R;                       // loads of one stage that go to the same region (contiguous there)
    unsigned off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = j * 4 + wave, region = q % R, slot = q / R;
        off[j] = (unsigned)(region * D + slot * 1024 + lane * 16);
    }
    const unsigned step = per_region * 1024;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* dst = smem + (((s & 3) * 16 + j * 4 + wave) * 256);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, (int)(off[j] + s * step), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (smem[threadIdx.x] == 123.456f) sink[1] = 1.f;
}

int main() {
    const size_t cap = 1536u << 20;
    float* src; long long* out; float* sink; unsigned char* junk;
    CK(hipMalloc(&src, cap)); CK(hipMemset(src, 0, cap));
    CK(hipMalloc(&junk, 512u << 20));
    CK(hipMalloc(&out, 8192 * 8)); CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int stages = 64;
    for (int nwg : {16, 160})
        for (int R : {1, 2, 4, 8, 16})
            for (long long D : {4096ll, 65536ll, 1ll << 20, 2ll << 20, 16ll << 20, 64ll << 20}) {
                if (R == 1 && D != 4096) continue;
                if ((long long)R * D + stages * 16384 > (long long)cap) continue;
                double best = 1e30;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(junk, rep, 512u << 20));
                    CK(hipDeviceSynchronize());
                    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 65536, 0, src, R, D, stages, out, sink);
                    CK(hipDeviceSynchronize());
                    std::vector<long long> h(nwg);
                    CK(hipMemcpy(h.data(), out, nwg * sizeof(long long), hipMemcpyDeviceToHost));
                    double cyc = 0; for (auto v : h) cyc += v; cyc /= nwg;
                    if (cyc < best) best = cyc;
                }
                printf("wgs %3d regions %2d spaced %8lld KiB: %6.0f clk per 16 KiB stage = %6.1f GB/s for the stream\n", nwg, R, D >> 10, best / stages,
                       16384.0 * stages / (best / 2.1e9) / 1e9);
            }
    return 0;
}
