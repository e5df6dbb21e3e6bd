// Minimal reproducer candidate for what the delta-debugging of the fused stem's assembly isolated (profiles/EXPERIMENTS.md, round 6):
//     v_pk_mul_f32 vD[0:1], vA[0:1], vB[0:1] op_sel:[0,1] op_sel_hi:[1,0]        (D.lo = A.lo * B.hi, D.hi = A.hi * B.lo)
// returns wrong products in some lanes while a kernel of ANOTHER stream issues v_mfma_f32_32x32x16_bf16 on the same SIMDs.
// Victim: every lane multiplies known pairs with the packed instruction (several operand-select forms) and with two scalar v_mul_f32, and counts
// the products that differ bit for bit.  Noise: a register-only MFMA loop (bf16 32x32x16 / fp32 32x32x2 / none) on a second stream.
//
// Build:  hipcc --offload-arch=gfx950 -O2 -o tools/stale_read/pkmul tools/stale_read/pkmul.hip
// Run:    tools/stale_read/pkmul [noise: 0 none | 1 bf16 32x32x16 MFMA | 2 fp32 MFMA | 3 plain VALU fma | 4 bf16 32x32x8 MFMA | 5 f16 32x32x16 MFMA | 6 bf16 16x16x32 MFMA | 7 fp8 32x32x16 MFMA] [form 0..9] [seconds]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// FORM 0: op_sel:[0,1] op_sel_hi:[1,0] (the instruction of the failing kernel)   1: no operand select (lo*lo, hi*hi)
//      2: op_sel_hi:[1,0] only (B.lo broadcast)                                   3: v_pk_fma_f32 with the cross select, C = 0
template <int FORM>
__global__ void __launch_bounds__(256) victim(unsigned* __restrict__ bad_per_lane, unsigned long long* __restrict__ first_bad, int iters, unsigned seed) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned s = seed ^ (tid * 2654435761u);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u; const float a0 = 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f);
        s = s * 1664525u + 1013904223u; const float a1 = 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f);
        s = s * 1664525u + 1013904223u; const float b0 = 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f);
        s = s * 1664525u + 1013904223u; const float b1 = 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f);
        f32x2 a = {a0, a1}, b = {b0, b1}, d;
        float e0, e1;
        if constexpr (FORM == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b1), "v"(a1), "v"(b0));
        } else if constexpr (FORM == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        } else if constexpr (FORM == 2) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b0), "v"(a1), "v"(b0));
        } else if constexpr (FORM == 3) {
            f32x2 z = {0.f, 0.f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(z));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b1), "v"(a1), "v"(b0));
        } else if constexpr (FORM == 4) {      // src0 cross: D.lo = A.hi * B.lo, D.hi = A.lo * B.hi
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a1), "v"(b0), "v"(a0), "v"(b1));
        } else if constexpr (FORM == 5) {      // packed add, src1 cross
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b1), "v"(a1), "v"(b0));
        } else if constexpr (FORM == 6) {      // low half takes B.hi, high half default (B.hi): both products with B.hi (warp_params_kernel's form, VGPR src0 here)
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b1), "v"(a1), "v"(b1));
        } else if constexpr (FORM == 7) {      // v_pk_mov_b32 D = {A.hi, B.lo}  (78 of these in the library)
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(d) : "v"(a), "v"(b));
            e0 = a1; e1 = b0;
        } else if constexpr (FORM == 8) {      // warp_inv_rot_norm_kernel's form: fma, src0 cross in the low half, SGPR src1
            const float u0 = 1.25f, u1 = 0.75f;
            f32x2 u = {u0, u1};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=&v"(d) : "v"(a), "s"(u), "v"(b));
            asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %6, %7" : "=&v"(e0), "=&v"(e1) : "v"(a1), "v"(u0), "v"(b0), "v"(a1), "v"(u1), "v"(b1));
        } else if constexpr (FORM == 10) {     // fma with the cross select on src2 (head_upsample_kernel's old form): D.lo = A.lo * B.lo + C.hi
            f32x2 c = {a1, b0};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %6, %7" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b0), "v"(b0), "v"(a1), "v"(b1), "v"(a1));
        } else {                               // 9: in place, as hipcc emitted it in the stem: D = B
            d = b;
            asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d) : "v"(a));
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(a0), "v"(b1), "v"(a1), "v"(b0));
        }
        const bool w0 = __float_as_uint(d.x) != __float_as_uint(e0), w1 = __float_as_uint(d.y) != __float_as_uint(e1);
        if (w0 || w1) {
            ++bad;
            if (bad == 1) atomicMin(first_bad, ((unsigned long long)it << 32) | ((unsigned long long)(threadIdx.x & 63) << 8) | (w0 ? 1u : 0u) | (w1 ? 2u : 0u));
        }
    }
    if (bad) atomicAdd(&bad_per_lane[threadIdx.x & 63], bad);
}

template <int MODE>
__global__ void __launch_bounds__(256) noise(float* __restrict__ sink, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4v fa = {1.f, 2.f, 3.f, 4.f}, fb = {0.5f, 0.25f, 0.125f, 1.f};
    float v = (float)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 1) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, fa), b = __builtin_bit_cast(bf16x8, fb);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        } else if constexpr (MODE == 4) {      // the 8-k bf16 MFMA of gfx940 (v_mfma_f32_32x32x8_bf16_1k)
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            const s16x4 a4 = {1, 2, 3, 4}, b4 = {5, 6, 7, 8};
            acc0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(b4, a4, acc1, 0, 0, 0);
        } else if constexpr (MODE == 5) {      // fp16 16-k MFMA (also new in gfx950)
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 a8 = __builtin_bit_cast(h8, fa), b8 = __builtin_bit_cast(h8, fb);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8, a8, acc1, 0, 0, 0);
        } else if constexpr (MODE == 6) {      // 16x16x32 bf16 (the other 16-k-class bf16 MFMA of gfx950)
            typedef float f32x4a __attribute__((ext_vector_type(4)));
            const bf16x8 a = __builtin_bit_cast(bf16x8, fa), b = __builtin_bit_cast(bf16x8, fb);
            f32x4a c0 = {acc0[0], acc0[1], acc0[2], acc0[3]}, c1 = {acc1[0], acc1[1], acc1[2], acc1[3]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
            acc0[0] = c0[0]; acc0[1] = c0[1]; acc0[2] = c0[2]; acc0[3] = c0[3]; acc1[0] = c1[0]; acc1[1] = c1[1]; acc1[2] = c1[2]; acc1[3] = c1[3];
        } else if constexpr (MODE == 7) {      // fp8 32x32x16 (gfx940 already has it)
            const long a8 = 0x3838383838383838L, b8 = 0x3C3C3C3C3C3C3C3CL;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b8, a8, acc1, 0, 0, 0);
        } else if constexpr (MODE == 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc1, 0, 0, 0);
        } else {
            v = fmaf(v, 1.0001f, 0.5f);
        }
    }
    float r = v;
    for (int k = 0; k < 16; ++k) r += acc0[k] + acc1[k];
    if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int FORM>
static void run(int nmode, double seconds) {
    hipStream_t sv, sn;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    unsigned* bad; unsigned long long* first; float* sink;
    CK(hipMalloc(&bad, 64 * 4)); CK(hipMalloc(&first, 8)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(bad, 0, 256)); CK(hipMemset(first, 0xFF, 8));
    hipEvent_t ev[2];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto nz = [&]() {
        if (nmode == 1) hipLaunchKernelGGL(noise<1>, dim3(2048), dim3(256), 0, sn, sink, 4000);
        else if (nmode == 2) hipLaunchKernelGGL(noise<2>, dim3(2048), dim3(256), 0, sn, sink, 4000);
        else if (nmode == 3) hipLaunchKernelGGL(noise<3>, dim3(2048), dim3(256), 0, sn, sink, 200000);
        else if (nmode == 4) hipLaunchKernelGGL(noise<4>, dim3(2048), dim3(256), 0, sn, sink, 4000);
        else if (nmode == 5) hipLaunchKernelGGL(noise<5>, dim3(2048), dim3(256), 0, sn, sink, 4000);
        else if (nmode == 6) hipLaunchKernelGGL(noise<6>, dim3(2048), dim3(256), 0, sn, sink, 8000);
        else if (nmode == 7) hipLaunchKernelGGL(noise<7>, dim3(2048), dim3(256), 0, sn, sink, 4000);
    };
    nz(); CK(hipEventRecord(ev[0], sn)); nz(); CK(hipEventRecord(ev[1], sn));
    int turn = 0;
    unsigned long long launches = 0;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventRecord(t0, sv));
    for (;;) {
        hipLaunchKernelGGL(victim<FORM>, dim3(1024), dim3(256), 0, sv, bad, first, 2000, (unsigned)launches * 7919u + 1u);
        ++launches;
        if (nmode && hipEventQuery(ev[turn]) == hipSuccess) { nz(); CK(hipEventRecord(ev[turn], sn)); turn ^= 1; }
        if ((launches & 15) == 0) {
            CK(hipStreamSynchronize(sv));
            CK(hipEventRecord(t1, sv)); CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (ms > seconds * 1e3) break;
        }
    }
    CK(hipDeviceSynchronize());
    unsigned hb[64]; unsigned long long hf;
    CK(hipMemcpy(hb, bad, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, first, 8, hipMemcpyDeviceToHost));
    unsigned long long tot = 0; unsigned q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += hb[l]; q[l >> 4] += hb[l]; }
    printf("PKMUL form=%d noise=%d: %llu victim launches x 262144 lanes x 2000 products: %llu wrong | by lane quarter [0-15] %u [16-31] %u [32-47] %u [48-63] %u", FORM, nmode, launches, tot,
           q[0], q[1], q[2], q[3]);
    if (tot) printf(" | earliest: iteration %llu lane %llu halves %llu", hf >> 32, (hf >> 8) & 0xFF, hf & 3);
    printf("\n");
}

int main(int argc, char** argv) {
    const int nmode = argc > 1 ? atoi(argv[1]) : 1, form = argc > 2 ? atoi(argv[2]) : 0;
    const double secs = argc > 3 ? atof(argv[3]) : 3.0;
    switch (form) {
        case 0: run<0>(nmode, secs); break; case 1: run<1>(nmode, secs); break; case 2: run<2>(nmode, secs); break; case 3: run<3>(nmode, secs); break;
        case 4: run<4>(nmode, secs); break; case 5: run<5>(nmode, secs); break; case 6: run<6>(nmode, secs); break; case 7: run<7>(nmode, secs); break;
        case 8: run<8>(nmode, secs); break; case 10: run<10>(nmode, secs); break; default: run<9>(nmode, secs); break;
    }
    return 0;
}
