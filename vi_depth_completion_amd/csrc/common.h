// Shared helpers for libvidc.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/vidc.h"

namespace vidc {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(vidc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define VIDC_REQUIRE(cond, code, ...)                  \
    do {                                               \
        if (!(cond)) {                                 \
            ::vidc::set_error(__VA_ARGS__);            \
            return (code);                             \
        }                                              \
    } while (0)

#define VIDC_CHECK_LAUNCH(name)                                                          \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::vidc::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return VIDC_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define VIDC_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ::vidc::set_error("%s failed: %s", #call, hipGetErrorString(e__));                \
            return VIDC_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- split-bf16 ("bf16x3") operand format -------------------------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi), round-to-nearest-even like torch's .to(bfloat16) (no NaN inputs here).
// A tensor with `ld` channels per row keeps every 32-channel unit in place as [32 x hi | 32 x lo] (the same 128 bytes).
__device__ inline unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float((unsigned)hi << 16));
}
// element (row, c) of a split image
__device__ inline void store_split(unsigned short* img, size_t row, int ld, int c, float v) {
    unsigned short hi, lo;
    split_bf16(v, hi, lo);
    unsigned short* u = img + (row * ld + (c & ~31)) * 2 + (c & 31);
    u[0] = hi;
    u[32] = lo;
}
// four consecutive channels c..c+3 (c % 4 == 0): two 8-byte stores
__device__ inline void store_split4(unsigned short* img, size_t row, int ld, int c, float4 v) {
    unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
    split_bf16(v.x, h0, l0); split_bf16(v.y, h1, l1); split_bf16(v.z, h2, l2); split_bf16(v.w, h3, l3);
    unsigned short* u = img + (row * ld + (c & ~31)) * 2 + (c & 31);
    *reinterpret_cast<uint2*>(u) = make_uint2(h0 | ((unsigned)h1 << 16), h2 | ((unsigned)h3 << 16));
    *reinterpret_cast<uint2*>(u + 32) = make_uint2(l0 | ((unsigned)l1 << 16), l2 | ((unsigned)l3 << 16));
}

}  // namespace vidc
