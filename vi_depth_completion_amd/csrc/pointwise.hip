// HBM-bound glue kernels of the two networks: stem conv (Cin 1/3), 3x3/s2 max-pool, bilinear align_corners=True
// upsample, and the prediction-head tail (1x1 conv with few outputs as a wavefront reduction + final upsample).
// All NHWC fp32 with 16-byte per-lane accesses where the layout allows.
#include "common.h"
#include <cstdlib>

namespace {

// ---- stem: 3x3 stride-2 pad-1 conv, Cin in {1,3}, NCHW in -> NHWC out (networks/surface_normal.py:17-18) --------
// One workgroup = 64 output pixels of one row segment x all Cout (<=64 per pass) channels.  The 27 (or 9) weights of
// a lane's output channel live in registers; the input patch is staged in LDS and broadcast.
// WARP = true (round 5): the conv's input is the gravity-aligned forward warp of x (warping_2dof_alignment.py:108-156 in front of
// surface_normal.py:163) and the patch loader gathers it on the fly -- one projective map and one bilinear tap set per patch pixel, shared
// by the CIN channels, the very code of warp_fwd_kernel (vidc::warp_fwd_taps / vidc::sample), so the patch holds the bits that kernel would
// have stored.  The warped image is never written: one launch and a 2 x 3HW x 4 byte round trip less per frame.
template <int CIN, bool WARP, bool CONTROL = false>
__global__ void __launch_bounds__(256)
stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int H, int W, int Ho, int Wo,
                 int Cout, int ldy, int relu, unsigned short* __restrict__ ysp, int ch0, const float* __restrict__ warp_params, float wcx, float wcy,
                 int align_corners) {
    constexpr int PIX = 64;                 // output pixels per workgroup (along X)
    constexpr int PW = 2 * PIX + 1;         // input patch width
    __shared__ float patch[CIN][3][PW + 3];
    const int b = blockIdx.z, oy = blockIdx.y, ox0 = blockIdx.x * PIX;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * CIN * plane;
    const int ix_base = ox0 * 2 - 1, iy_base = oy * 2 - 1;
    if constexpr (WARP) {
        // The record (written by warp_params_kernel, the launch before this one) goes through vector loads into LDS: the tap code wants the 32 floats
        // as lane-indexed data.  (Round 5 blamed wave-uniform s_load reads of it for wrong frames; round 6 measured that form innocent)
        __shared__ float wp[VIDC_WARP_PARAMS];
        if (tid < VIDC_WARP_PARAMS) wp[tid] = __builtin_nontemporal_load(warp_params + (size_t)b * VIDC_WARP_PARAMS + tid);
        __syncthreads();
        for (int e = tid; e < 3 * PW; e += 256) {
            const int r = e / PW, i = e - r * PW;
            const int iy = iy_base + r, ix = ix_base + i;
            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;      // (outside: the conv's zero padding, not a warp sample)
            vidc::Taps t;
            if (in) t = vidc::warp_fwd_taps(wp, ix, iy, wcx, wcy, W, H, align_corners);
            if constexpr (CONTROL) {
                // THE POSITIVE CONTROL of tests/test_stale_reads.py (VIDC_DBG_STEM_LOADS=3).  Round 6 delta-debugged hipcc's assembly of this kernel's then
                // plain-load form down to one instruction: `v_pk_mul_f32 v[28:29], v[36:37], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]` (the two cross bilinear
                // weights in one packed multiply).  On MI355X that encoding -- a packed-fp32 op whose LOW half takes the HIGH dword of src1 -- returns a
                // wrong low half in lanes 48-63 while another wave of the SIMD issues v_mfma_f32_32x32x16_bf16 / _f16 (tools/stale_read/pkmul.hip: 100 lines,
                // no memory involved).  This file is built without SLP-vectorised packed fp32 (csrc/Makefile, tools/audit_isa.py), so the debug
                // form puts the instruction back ON PURPOSE, as an identity multiply of the two weights: correct hardware leaves them unchanged.
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                f32x2_ wa = {t.w01, t.w10}, ones = {1.0f, 1.0f}, wd;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(wd) : "v"(wa), "v"(ones));
                t.w01 = wd.x;
                t.w10 = wd.y;
            }
            // (system-scope tap loads: a leftover of round 5's stale-line theory, harmless -- 3 MB per launch; the other load flavours and the hand-written
            //  schedules of round 6's bisect lived here until commit 5f9fbb5 and are described in profiles/EXPERIMENTS.md)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                float v = 0.f;
                if (in) {
                    const float* pl = xb + c * plane;
                    const float a00 = __hip_atomic_load(pl + t.o00, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), a01 = __hip_atomic_load(pl + t.o01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const float a10 = __hip_atomic_load(pl + t.o10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), a11 = __hip_atomic_load(pl + t.o11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    v = fmaf(a11, t.w11, fmaf(a10, t.w10, fmaf(a01, t.w01, a00 * t.w00)));
                }
                patch[c][r][i] = v;
            }
        }
    } else {
        for (int e = tid; e < CIN * 3 * PW; e += 256) {
            int c = e / (3 * PW), r = (e / PW) % 3, i = e % PW;
            int iy = iy_base + r, ix = ix_base + i;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = __builtin_nontemporal_load(xb + c * plane + (size_t)iy * W + ix);      // (rewritten every tick: common.h sample_nt)
            patch[c][r][i] = v;
        }
    }
    __syncthreads();
    const int co = tid & 63, pgrp = tid >> 6;      // lanes run along channels -> 256-byte coalesced NHWC stores
    for (int cbase = 0; cbase < Cout; cbase += 64) {
        const int c_out = cbase + co;
        float wr[CIN * 9];
        if (c_out < Cout) {
#pragma unroll
            for (int k = 0; k < CIN * 9; ++k) wr[k] = w[(size_t)c_out * CIN * 9 + k];
        }
        for (int p = pgrp; p < PIX; p += 4) {
            const int ox = ox0 + p;
            if (ox >= Wo || c_out >= Cout) continue;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) acc += patch[c][r][2 * p + s] * wr[(c * 3 + r) * 3 + s];
            if (relu) acc = fmaxf(acc, 0.f);
            const size_t row = (size_t)(b * Ho + oy) * Wo + ox;
            if (y) y[row * ldy + c_out] = acc;
            if (ysp) vidc::store_split(ysp, row, ldy, ch0 + c_out, acc);      // ysp is the image of the whole [.., ldy] tensor
        }
    }
}

// ---- max-pool 3x3 stride 2 pad 1, NHWC -----------------------------------------------------------------------
// Workgroups are handed to the 8 XCDs (8 private L2s) round-robin in linear launch order.  These kernels read every input row from
// several output rows and every input pixel from neighbouring column blocks, so instead of letting neighbours land on different
// XCDs, XCD k takes the k-th contiguous BAND of the (row, column-block) space: workgroup L of the launch = the (L / 8)-th block of XCD
// L % 8.  gridDim.x is a multiple of 8 (host side), so every XCD gets the same number of blocks.  (rocprofv3 FETCH_SIZE: the bilinear
// upsample fetched 3.3-6.6x its input before, the max-pool 1.6x.)
__device__ __forceinline__ void xcd_band_block(unsigned& bx, unsigned& by) {
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, per = (gx >> 3) * gridDim.y;
    const unsigned idx = (L & 7u) * per + (L >> 3);
    by = idx / gx;
    bx = idx - by * gx;
}
// The same idea with the bands cut into groups of kRowGroup rows dealt to the XCDs in turn (gridDim.y a multiple of 8 * kRowGroup): the
// eight L2s then work within a window of 8 * kRowGroup rows of the output instead of in eight regions far apart, which keeps the DRAM
// pages they write close together; only the rows at the group seams are fetched twice.
constexpr unsigned kRowGroup = 4;
__device__ __forceinline__ void xcd_rowgroup_block(unsigned& bx, unsigned& by) {
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x;
    const unsigned k = L & 7u, j = L >> 3, per_group = kRowGroup * gx;
    const unsigned t = j / per_group, r = j - t * per_group;
    const unsigned rr = r / gx;
    by = (t * 8u + k) * kRowGroup + rr;
    bx = r - rr * gx;
}

__global__ void __launch_bounds__(256)
maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int ldx, int Ho, int Wo, int ldy,
               unsigned short* __restrict__ ysp) {
    // grid.y = output row (image, oy): wave-uniform, so the per-thread index math is one 32-bit division (the first version
    // decomposed a 64-bit linear index with three emulated 64-bit divisions per thread: VALU-bound at 52-62 % of the HBM peak)
    const unsigned q = (unsigned)C / 4u;
    unsigned bx, by;
    xcd_band_block(bx, by);
    const unsigned i = bx * blockDim.x + threadIdx.x;
    if (i >= (unsigned)Wo * q) return;
    const int ox = (int)(i / q);
    const int c = (int)(i - (unsigned)ox * q) * 4;
    const int b = (int)(by / (unsigned)Ho), oy = (int)(by - (unsigned)b * (unsigned)Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int iy = oy * 2 - 1 + r;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            int ix = ox * 2 - 1 + s;
            if ((unsigned)ix >= (unsigned)W) continue;
            float4 v = *reinterpret_cast<const float4*>(&x[((size_t)(b * H + iy) * W + ix) * ldx + c]);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    const size_t row = (size_t)(b * Ho + oy) * Wo + ox;
    if (y) *reinterpret_cast<float4*>(&y[row * ldy + c]) = m;
    if (ysp) vidc::store_split4(ysp, row, ldy, c, m);
}

// ---- bilinear upsample, align_corners=True (nn.UpsamplingBilinear2d), NHWC ---------------------------------------
// ATen's area_pixel_compute_source_index(scale=(in-1)/(out-1), dst, align_corners=true) = scale*dst.
__device__ inline void src_index(int dst, int in, int out, int& i0, int& i1, float& l1) {
    float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float s = scale * (float)dst;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ void __launch_bounds__(256)
upsample_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int h, int w, int C, int ldx, int H, int W, int ldy,
                int flags, unsigned short* __restrict__ ysp) {
    const unsigned q = (unsigned)C / 4u;                   // grid.y = output row (image, oy), see maxpool_kernel
    unsigned bx, by;
    xcd_rowgroup_block(bx, by);
    const unsigned i = bx * blockDim.x + threadIdx.x;
    if (i >= (unsigned)W * q) return;
    const int ox = (int)(i / q);
    const int c = (int)(i - (unsigned)ox * q) * 4;
    if (by >= (unsigned)(B * H)) return;                   // rows of the padded grid
    const int b = (int)(by / (unsigned)H), oy = (int)(by - (unsigned)b * (unsigned)H);
    int y0, y1, x0, x1; float ly, lx;
    src_index(oy, h, H, y0, y1, ly);
    src_index(ox, w, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const size_t row = (size_t)(b * H + oy) * W + ox;
    float4* dst = reinterpret_cast<float4*>(&y[row * ldy + c]);
    const int G = (flags >> 8) & 0xFF;                 // > 1: x holds G groups of C channels whose upsampled values are summed, in order
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (flags & VIDC_UP_ACCUM) o = *dst;
    for (int g = 0; g < (G > 1 ? G : 1); ++g) {
        const float* xb = x + (size_t)b * h * w * ldx + g * C + c;
        const float4 v00 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * w + x0) * ldx);
        const float4 v01 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * w + x1) * ldx);
        const float4 v10 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * w + x0) * ldx);
        const float4 v11 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * w + x1) * ldx);
        float4 u;
        u.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        u.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        u.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        u.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        if (flags & VIDC_UP_RELU) { u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f); }
        o.x = u.x + o.x; o.y = u.y + o.y; o.z = u.z + o.z; o.w = u.w + o.w;       // value + running sum, like one launch per group would
    }
    if (!(flags & VIDC_UP_NO_F32_OUT)) *dst = o;
    if (ysp) vidc::store_split4(ysp, row, ldy, c, o);
}

// ---- head: 1x1 conv to <=4 channels with zero padding, one wavefront per low-res pixel ------------------------------
// (the "wavefront-reduction" part of the path: each lane owns Cin/64 channels, DPP/shuffle tree over 64 lanes)
__global__ void __launch_bounds__(256)
head_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ low,
                 int B, int h, int wd, int Cin, int ldx, int Cout, int pad) {
    const int hp = h + 2 * pad, wp = wd + 2 * pad;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int total = B * hp * wp;
    if (wave >= total) return;
    const int px = wave % wp, py = (wave / wp) % hp, b = wave / (wp * hp);
    const int iy = py - pad, ix = px - pad;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)wd) {
        const float* xp = x + ((size_t)(b * h + iy) * wd + ix) * ldx;
        for (int c = lane; c < Cin; c += 64) {
            float v = xp[c];
            for (int o = 0; o < Cout; ++o) acc[o] += v * w[o * Cin + c];
        }
    }
    for (int o = 0; o < Cout; ++o) {
        float v = acc[o];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) low[((size_t)(b * Cout + o) * hp + py) * wp + px] = v + bias[o];
    }
}

// final UpsamplingBilinear2d(size=(H,W)) of the small NCHW map, optional ReLU, NCHW out (coalesced along X)
__global__ void __launch_bounds__(256)
head_upsample_kernel(const float* __restrict__ low, float* __restrict__ y, int BC, int h, int w, int H, int W, int relu) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)BC * H * W) return;
    const int ox = (int)(idx % W);
    const int oy = (int)((idx / W) % H);
    const int bc = (int)(idx / ((long long)W * H));
    int y0, y1, x0, x1; float ly, lx;
    src_index(oy, h, H, y0, y1, ly);
    src_index(ox, w, W, x0, x1, lx);
    const float* p = low + (size_t)bc * h * w;
    const float hy = 1.f - ly, hx = 1.f - lx;
    float v = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
    if (relu) v = fmaxf(v, 0.f);
    y[idx] = v;
}

// ---- nn.AvgPool2d(k, stride, padding) with count_include_pad=True (FullImageEncoder, surface_normal_dorn.py:10), NHWC -------------
__global__ void __launch_bounds__(256)
avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int ldx, int kh, int kw, int sh, int sw, int ph,
               int pw, int Ho, int Wo, int ldy) {
    const int q = C / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * Ho * Wo * q) return;
    const int c = (int)(idx % q) * 4;
    long long t = idx / q;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < kh; ++r) {
        const int iy = oy * sh - ph + r;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int u = 0; u < kw; ++u) {
            const int ix = ox * sw - pw + u;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(&x[((size_t)(b * H + iy) * W + ix) * ldx + c]);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    const float d = (float)(kh * kw);                       // padding counts
    *reinterpret_cast<float4*>(&y[((size_t)(b * Ho + oy) * Wo + ox) * ldy + c]) = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

// ---- F.normalize(x, dim=1) on NCHW (surface_normal_dorn.py:154): x / max(||x||_2, 1e-12) per pixel --------------------------------
__global__ void __launch_bounds__(256) normalize_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * HW) return;
    const int b = (int)(idx / HW), p = (int)(idx - (long long)b * HW);
    const float* xb = x + (size_t)b * C * HW + p;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float v = xb[(size_t)c * HW]; ss += v * v; }
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    float* yb = y + (size_t)b * C * HW + p;
    for (int c = 0; c < C; ++c) yb[(size_t)c * HW] = xb[(size_t)c * HW] * inv;
}

}  // namespace

extern "C" int vidc_avgpool2d(const float* x, float* y, int B, int H, int W, int C, int ldx, int kh, int kw, int sh, int sw, int ph, int pw,
                              int ldy, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y, VIDC_ERR_NULL, "vidc_avgpool2d: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 &&
                     pw >= 0 && H + 2 * ph >= kh && W + 2 * pw >= kw,
                 VIDC_ERR_SHAPE, "vidc_avgpool2d: bad shape");
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), x, y, B, H, W, C, ldx, kh, kw, sh,
                       sw, ph, pw, Ho, Wo, ldy);
    VIDC_CHECK_LAUNCH("avgpool_kernel");
    return VIDC_OK;
}

extern "C" int vidc_normalize_nchw(const float* x, float* y, int B, int C, int HW, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y, VIDC_ERR_NULL, "vidc_normalize_nchw: null pointer");
    VIDC_REQUIRE(B > 0 && C > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_normalize_nchw: bad shape");
    const long long total = (long long)B * HW;
    hipLaunchKernelGGL(normalize_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), x, y, B, C, HW);
    VIDC_CHECK_LAUNCH("normalize_nchw_kernel");
    return VIDC_OK;
}

namespace {
int stem_launch(const float* x, const float* w_oihw, float* y, int B, int Cin, int H, int W, int Cout, int ldy, int relu, void* y_split, int split_ch0,
                const float* warp_params, float cx, float cy, int align_corners, vidc_stream_t stream);
}

extern "C" int vidc_stem_conv3x3s2(const float* x, const float* w_oihw, float* y, int B, int Cin, int H, int W, int Cout, int ldy,
                                   int relu, void* y_split, int split_ch0, vidc_stream_t stream) {
    return stem_launch(x, w_oihw, y, B, Cin, H, W, Cout, ldy, relu, y_split, split_ch0, nullptr, 0.f, 0.f, 0, stream);
}

extern "C" int vidc_stem_conv3x3s2_warped(const float* x, const float* warp_params, const float* w_oihw, float* y, int B, int H, int W, int Cout, int ldy,
                                          int relu, void* y_split, int split_ch0, float cx, float cy, int align_corners, vidc_stream_t stream) {
    VIDC_REQUIRE(warp_params, VIDC_ERR_NULL, "vidc_stem_conv3x3s2_warped: null warp parameters");
    return stem_launch(x, w_oihw, y, B, 3, H, W, Cout, ldy, relu, y_split, split_ch0, warp_params, cx, cy, align_corners, stream);
}

namespace {
int stem_launch(const float* x, const float* w_oihw, float* y, int B, int Cin, int H, int W, int Cout, int ldy, int relu, void* y_split, int split_ch0,
                const float* warp_params, float cx, float cy, int align_corners, vidc_stream_t stream) {
    VIDC_REQUIRE(x && w_oihw && (y || y_split), VIDC_ERR_NULL, "vidc_stem_conv3x3s2: null pointer");
    VIDC_REQUIRE(!y_split || (ldy % 32 == 0 && split_ch0 >= 0), VIDC_ERR_SHAPE, "vidc_stem_conv3x3s2: split output needs ldy % 32 == 0");
    unsigned short* ysp = reinterpret_cast<unsigned short*>(y_split);
    VIDC_REQUIRE((Cin == 1 || Cin == 3) && B > 0 && H > 2 && W > 2 && Cout > 0 && ldy >= Cout, VIDC_ERR_SHAPE,
                 "vidc_stem_conv3x3s2: unsupported shape (Cin=%d must be 1 or 3)", Cin);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    dim3 grid(vidc::cdiv(Wo, 64), Ho, B);
    hipStream_t st = vidc::as_stream(stream);
    // VIDC_DBG_STEM_LOADS=3 (test control only): the form that carries the defective packed multiply on purpose (see the kernel)
    static const int dbg_loads = [] { const char* e = getenv("VIDC_DBG_STEM_LOADS"); return e ? atoi(e) : 0; }();
    if (warp_params && dbg_loads == 3)
        hipLaunchKernelGGL((stem_conv_kernel<3, true, true>), grid, dim3(256), 0, st, x, w_oihw, y, H, W, Ho, Wo, Cout, ldy, relu, ysp, split_ch0, warp_params, cx, cy, align_corners);
    else if (warp_params)
        hipLaunchKernelGGL((stem_conv_kernel<3, true>), grid, dim3(256), 0, st, x, w_oihw, y, H, W, Ho, Wo, Cout, ldy, relu, ysp, split_ch0, warp_params, cx, cy, align_corners);
    else if (Cin == 3)
        hipLaunchKernelGGL((stem_conv_kernel<3, false>), grid, dim3(256), 0, st, x, w_oihw, y, H, W, Ho, Wo, Cout, ldy, relu, ysp, split_ch0, warp_params, cx, cy, align_corners);
    else
        hipLaunchKernelGGL((stem_conv_kernel<1, false>), grid, dim3(256), 0, st, x, w_oihw, y, H, W, Ho, Wo, Cout, ldy, relu, ysp, split_ch0, warp_params, cx, cy, align_corners);
    VIDC_CHECK_LAUNCH("stem_conv_kernel");
    return VIDC_OK;
}
}  // namespace

extern "C" int vidc_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, void* y_split,
                                 vidc_stream_t stream) {
    VIDC_REQUIRE(x && (y || y_split), VIDC_ERR_NULL, "vidc_maxpool3x3s2: null pointer");
    VIDC_REQUIRE(!y_split || ldy % 32 == 0, VIDC_ERR_SHAPE, "vidc_maxpool3x3s2: split output needs ldy % 32 == 0");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C,
                 VIDC_ERR_SHAPE, "vidc_maxpool3x3s2: bad shape (C, ldx, ldy must be multiples of 4)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    VIDC_REQUIRE((long long)B * Ho <= 65535, VIDC_ERR_SHAPE, "vidc_maxpool3x3s2: B * Ho = %lld rows exceed the grid", (long long)B * Ho);
    // grid.x rounded up to a multiple of 8 (idle workgroups return at once) for xcd_band_block()
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)(vidc::cdiv(Wo * (C / 4), 256) + 7) / 8 * 8, (unsigned)(B * Ho)), dim3(256), 0, vidc::as_stream(stream), x, y, B, H, W,
                       C, ldx, Ho, Wo, ldy, reinterpret_cast<unsigned short*>(y_split));
    VIDC_CHECK_LAUNCH("maxpool_kernel");
    return VIDC_OK;
}

extern "C" int vidc_upsample_bilinear_ac(const float* x, float* y, int B, int h, int w, int C, int ldx, int H, int W, int ldy,
                                         int flags, void* y_split, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y, VIDC_ERR_NULL, "vidc_upsample_bilinear_ac: null pointer");
    VIDC_REQUIRE(!y_split || ldy % 32 == 0, VIDC_ERR_SHAPE, "vidc_upsample_bilinear_ac: split output needs ldy % 32 == 0");
    VIDC_REQUIRE(!(flags & VIDC_UP_NO_F32_OUT) || y_split, VIDC_ERR_SHAPE, "vidc_upsample_bilinear_ac: NO_F32_OUT without y_split writes nothing");
    VIDC_REQUIRE(ldx >= C * (((flags >> 8) & 0xFF) > 1 ? ((flags >> 8) & 0xFF) : 1) && ldy >= C, VIDC_ERR_SHAPE, "vidc_upsample_bilinear_ac: bad channel strides");
    VIDC_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, VIDC_ERR_SHAPE,
                 "vidc_upsample_bilinear_ac: bad shape");
    VIDC_REQUIRE((long long)B * H <= 65504, VIDC_ERR_SHAPE, "vidc_upsample_bilinear_ac: B * H = %lld rows exceed the grid", (long long)B * H);
    // (grid.x a multiple of 8, grid.y of 32: xcd_rowgroup_block())
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)(vidc::cdiv(W * (C / 4), 256) + 7) / 8 * 8, (unsigned)((B * H + 31) / 32 * 32)), dim3(256), 0, vidc::as_stream(stream), x, y, B, h, w,
                       C, ldx, H, W, ldy, flags, reinterpret_cast<unsigned short*>(y_split));
    VIDC_CHECK_LAUNCH("upsample_kernel");
    return VIDC_OK;
}

extern "C" int vidc_head_conv1x1_upsample(const float* x, const float* w, const float* bias, float* lowres, float* y, int B, int h,
                                          int w_in, int Cin, int ldx, int Cout, int pad, int H, int W, int relu,
                                          vidc_stream_t stream) {
    VIDC_REQUIRE(x && w && bias && lowres && y, VIDC_ERR_NULL, "vidc_head_conv1x1_upsample: null pointer");
    VIDC_REQUIRE(B > 0 && h > 0 && w_in > 0 && Cin > 0 && Cout >= 1 && Cout <= 4 && pad >= 0 && H > 0 && W > 0, VIDC_ERR_SHAPE,
                 "vidc_head_conv1x1_upsample: bad shape (Cout must be 1..4)");
    const int hp = h + 2 * pad, wp = w_in + 2 * pad;
    const int waves = B * hp * wp;
    hipLaunchKernelGGL(head_conv_kernel, dim3(vidc::cdiv(waves, 4)), dim3(256), 0, vidc::as_stream(stream), x, w, bias, lowres, B, h,
                       w_in, Cin, ldx, Cout, pad);
    VIDC_CHECK_LAUNCH("head_conv_kernel");
    long long total = (long long)B * Cout * H * W;
    hipLaunchKernelGGL(head_upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), lowres, y,
                       B * Cout, hp, wp, H, W, relu);
    VIDC_CHECK_LAUNCH("head_upsample_kernel");
    return VIDC_OK;
}

// ---- use_mask branch of SurfaceNormalPrediction.forward (networks/surface_normal.py:150-162) ---------------------------------------
// feature_mask = (r + g + b > 1e-2) of the WARPED image, nearest-resized to the feature map (F.interpolate(mode='nearest'):
// src = min(floor(dst * in / out), in - 1) with the scale in fp32), multiplied into the features.  One launch: the mask is never stored.
namespace {
__global__ void __launch_bounds__(256)
mask_scale_kernel(const float* __restrict__ x, const float* __restrict__ img, float* __restrict__ y, int B, int h, int w, int C, int ldx, int ldy,
                  int H, int W) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * h * w * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    long long p = i / c4;
    const int xx = (int)(p % w); p /= w;
    const int yy = (int)(p % h);
    const int b = (int)(p / h);
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    const int iy = min((int)floorf((float)yy * sy), H - 1), ix = min((int)floorf((float)xx * sx), W - 1);
    const float* ib = img + (size_t)b * 3 * H * W + (size_t)iy * W + ix;
    const float m = (ib[0] + ib[(size_t)H * W] + ib[2 * (size_t)H * W] > 1e-2f) ? 1.f : 0.f;
    const size_t row = ((size_t)b * h + yy) * w + xx;
    const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
    *reinterpret_cast<float4*>(y + row * ldy + c) = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
}
}  // namespace

extern "C" int vidc_mask_scale(const float* x, const float* image_nchw, float* y, int B, int h, int w, int C, int ldx, int ldy, int H, int W,
                               vidc_stream_t stream) {
    VIDC_REQUIRE(x && image_nchw && y, VIDC_ERR_NULL, "vidc_mask_scale: null pointer");
    VIDC_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0, VIDC_ERR_SHAPE,
                 "vidc_mask_scale: bad shape");
    const long long total = (long long)B * h * w * (C / 4);
    hipLaunchKernelGGL(mask_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), x, image_nchw, y, B, h, w, C, ldx, ldy,
                       H, W);
    VIDC_CHECK_LAUNCH("mask_scale_kernel");
    return VIDC_OK;
}
