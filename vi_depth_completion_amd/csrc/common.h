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

// csrc/wgemm.hip: the streamed grouped GEMM behind vidc_conv2d_bn_act's tile VIDC_TILE_G96x32_STREAM
int launch_wgemm_stream(const vidc_conv_desc& d, hipStream_t st);
// csrc/wfused.hip: Winograd F(4x4, 3x3) in one launch behind the tile VIDC_TILE_WINO4_FUSED
int launch_wino4_fused(const vidc_conv_desc& d, hipStream_t st);

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

// ---- bilinear tap set, zero padding (grid_sampler_2d semantics) ---------------------------------------------
struct Taps {
    int o00, o01, o10, o11;       // element offsets inside one H*W plane (clamped, valid even if weight==0)
    float w00, w01, w10, w11;
};

__device__ inline Taps make_taps(float u, float v, float cx, float cy, int W, int H, int align_corners) {
    float gx = (1.0f / ((float)W / 2)) * (u - cx);
    float gy = (1.0f / ((float)H / 2)) * (v - cy);
    float ix, iy;
    if (align_corners) {
        ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
        iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    } else {
        ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
        iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    }
    // keep NaN/inf and far-out coordinates harmless: they sample nothing
    if (!(ix > -2.0f && ix < (float)W + 1.0f)) ix = -2.0f;
    if (!(iy > -2.0f && iy < (float)H + 1.0f)) iy = -2.0f;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix;
    float wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
    bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1), cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    Taps t;
    t.o00 = cy0 * W + cx0; t.o01 = cy0 * W + cx1; t.o10 = cy1 * W + cx0; t.o11 = cy1 * W + cx1;
    t.w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f;
    t.w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
    t.w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f;
    t.w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
    return t;
}

__device__ inline float sample(const float* __restrict__ plane, const Taps& t) {
    // explicit fma chain: the result must not depend on how the compiler unrolls the channel loop
    return fmaf(plane[t.o11], t.w11, fmaf(plane[t.o10], t.w10, fmaf(plane[t.o01], t.w01, plane[t.o00] * t.w00)));
}
// The same taps through non-temporal loads (global_load ... nt: streamed past the CU's vector L1).  For tensors that a copy or another kernel of the
// same tick rewrote just before this launch -- the frame, the normals, the enriched depth: every pixel is read by one or two workgroups, so there is
// nothing for the L1 to keep.  (Introduced in round 6 while the round-5 hazard was still read as a cache effect; its cause turned out to be an execution
// defect of packed fp32 ops, DESIGN 4.5 -- the loads stay because they are the right form for single-use data.)  Same bits as sample().
__device__ inline float sample_nt(const float* __restrict__ plane, const Taps& t) {
    const float a00 = __builtin_nontemporal_load(plane + t.o00), a01 = __builtin_nontemporal_load(plane + t.o01);
    const float a10 = __builtin_nontemporal_load(plane + t.o10), a11 = __builtin_nontemporal_load(plane + t.o11);
    return fmaf(a11, t.w11, fmaf(a10, t.w10, fmaf(a01, t.w01, a00 * t.w00)));
}

// Tap set of output pixel (X, Y) of the gravity-aligned forward warp (warping_2dof_alignment.py:142-152): H^-1 (X / kw + px_min, Y / kh + py_min, 1)
// from the per-sample record of vidc_warp2dof_params.  One definition for the stand-alone warp kernel (csrc/warp.hip) and for the stem conv
// that gathers its input through the warp (csrc/pointwise.hip): the same bits either way.
__device__ inline Taps warp_fwd_taps(const float* __restrict__ p, int X, int Y, float cx, float cy, int W, int H, int align_corners) {
    const float px_min = p[27], py_min = p[28], kw = p[29], kh = p[30];
    float Xs, Ys, P0, P1, P2;
    {
#pragma clang fp contract(off)
        Xs = (1.0f / kw) * (float)X + px_min;
        Ys = (1.0f / kh) * (float)Y + py_min;
        P0 = (p[18] * Xs + p[19] * Ys) + p[20];
        P1 = (p[21] * Xs + p[22] * Ys) + p[23];
        P2 = (p[24] * Xs + p[25] * Ys) + p[26];
    }
    return make_taps(P0 / P2, P1 / P2, cx, cy, W, H, align_corners);
}

}  // namespace vidc
