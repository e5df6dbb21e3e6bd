// 2-DoF gravity-aligned homography warp for gfx950 (HBM-bound bilinear gather).
//
// Replaces networks/warping_2dof_alignment.py:35-58,108-156,216-255 of the reference: there the sampling grid is
// materialised through ~520 ATen calls per sample; here a 1-thread-per-sample prologue derives (H, R, H^-1, bbox,
// kw, kh) and one gather kernel maps each output pixel through the homography and samples, so the only HBM traffic
// is the image in and the image out (1.84 MB per 3x240x320 frame).
#include "common.h"

namespace {

// ---- per-sample geometry ---------------------------------------------------------------------------------
__device__ inline void mat3_mul(const float* A, const float* B, float* C) {
#pragma clang fp contract(off)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = A[i * 3 + 0] * B[0 * 3 + j];
            s = s + A[i * 3 + 1] * B[1 * 3 + j];
            s = s + A[i * 3 + 2] * B[2 * 3 + j];
            C[i * 3 + j] = s;
        }
}

__global__ void warp_params_kernel(const float* __restrict__ gravity, const float* __restrict__ aligned, int B, float fx,
                                   float fy, float cx, float cy, const float* __restrict__ Kinv_in, int W, int H,
                                   float* __restrict__ params) {
#pragma clang fp contract(off)
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float g0 = gravity[b * 3 + 0], g1 = gravity[b * 3 + 1], g2 = gravity[b * 3 + 2];
    const float a0 = aligned[b * 3 + 0], a1 = aligned[b * 3 + 1], a2 = aligned[b * 3 + 2];
    // q = (-skew(a)) g = g x a ; rows of -skew(a): [0, a2, -a1], [-a2, 0, a0], [a1, -a0, 0]
    float q0 = a2 * g1 + (-a1) * g2;
    float q1 = (-a2) * g0 + a0 * g2;
    float q2 = a1 * g0 + (-a0) * g1;
    float dot = a0 * g0 + a1 * g1 + a2 * g2;
    float nq = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
    float q4 = cosf(0.5f * atan2f(nq, dot));
    // (the reference's degenerate-rotation branch, :49-50, is dead: its result is overwritten at :53)
    float d = 2.0f * q4;
    q0 = q0 / d; q1 = q1 / d; q2 = q2 / d;
    float S[9] = {0.f, -q2, q1, q2, 0.f, -q0, -q1, q0, 0.f};
    float S2[9], twoS[9];
    for (int i = 0; i < 9; ++i) twoS[i] = 2.0f * S[i];
    mat3_mul(twoS, S, S2);
    float R[9];
    for (int i = 0; i < 9; ++i) {
        float id = (i % 4 == 0) ? 1.0f : 0.0f;
        R[i] = (id + 2.0f * q4 * S[i]) + S2[i];
    }
    float K[9] = {fx, 0.f, cx, 0.f, fy, cy, 0.f, 0.f, 1.f};
    float Kinv[9], Rt[9], T[9], Hm[9], Hinv[9];
    for (int i = 0; i < 9; ++i) Kinv[i] = Kinv_in[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = R[j * 3 + i];
    mat3_mul(K, R, T);
    mat3_mul(T, Kinv, Hm);
    mat3_mul(K, Rt, T);
    mat3_mul(T, Kinv, Hinv);
    // bbox of the four projected image corners
    const float cxs[4] = {0.f, (float)(W - 1), 0.f, (float)(W - 1)};
    const float cys[4] = {0.f, 0.f, (float)(H - 1), (float)(H - 1)};
    float px_min = 0.f, px_max = 0.f, py_min = 0.f, py_max = 0.f;
    for (int c = 0; c < 4; ++c) {
        float x = (Hm[0] * cxs[c] + Hm[1] * cys[c]) + Hm[2];
        float y = (Hm[3] * cxs[c] + Hm[4] * cys[c]) + Hm[5];
        float z = (Hm[6] * cxs[c] + Hm[7] * cys[c]) + Hm[8];
        float px = x / z, py = y / z;
        if (c == 0) { px_min = px_max = px; py_min = py_max = py; }
        else {
            px_min = fminf(px_min, px); px_max = fmaxf(px_max, px);
            py_min = fminf(py_min, py); py_max = fmaxf(py_max, py);
        }
    }
    float h_max = py_max - py_min, w_max = px_max - px_min;
    float kw, kh;
    if (w_max > 4.0f * h_max / 3.0f) {
        kw = (float)W / w_max;
        kh = (float)H / (3.0f * w_max / 4.0f);
    } else {
        kh = (float)H / h_max;
        kw = (float)W / (4.0f * h_max / 3.0f);
    }
    float* p = params + (size_t)b * VIDC_WARP_PARAMS;
    for (int i = 0; i < 9; ++i) { p[i] = Hm[i]; p[9 + i] = R[i]; p[18 + i] = Hinv[i]; }
    p[27] = px_min; p[28] = py_min; p[29] = kw; p[30] = kh; p[31] = 0.f;
}

using vidc::Taps;
using vidc::make_taps;
using vidc::sample;
using vidc::warp_fwd_taps;

// Workgroups go to the 8 XCDs (8 private L2s) round-robin in linear launch order; a block of 256 consecutive output pixels gathers from
// source rows that its neighbours need too.  XCD k therefore takes the k-th contiguous band of the (image, pixel block) space instead of
// every eighth block (gridDim.x is a multiple of 8, host side): rocprofv3 FETCH_SIZE was 3.7x the input with the round-robin order.
__device__ __forceinline__ void xcd_band_block(unsigned& bx, unsigned& by) {
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, per = (gx >> 3) * gridDim.y;
    const unsigned idx = (L & 7u) * per + (L >> 3);
    by = idx / gx;
    bx = idx - by * gx;
}

// One thread per output pixel (lanes run along X, so the NCHW stores are fully coalesced and the
// gathers of neighbouring lanes hit neighbouring source pixels); all C channels reuse one tap set.
__global__ void __launch_bounds__(256)
warp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ params, float* __restrict__ y, int C, int H, int W,
                float cx, float cy, int align_corners) {
    unsigned bx, by;
    xcd_band_block(bx, by);
    const int b = (int)by;
    const int pix = (int)(bx * blockDim.x + threadIdx.x);
    if (pix >= H * W) return;
    const int Y = pix / W, X = pix - Y * W;
    const Taps t = warp_fwd_taps(params + (size_t)b * VIDC_WARP_PARAMS, X, Y, cx, cy, W, H, align_corners);
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * C * plane;
    float* yb = y + (size_t)b * C * plane + pix;
    for (int c = 0; c < C; ++c) yb[c * plane] = vidc::sample_nt(xb + c * plane, t);      // (x: rewritten by a copy before every launch -- common.h)
}

__global__ void __launch_bounds__(256)
warp_inv_rot_norm_kernel(const float* __restrict__ x, const float* __restrict__ params, float* __restrict__ z, int H, int W,
                         float cx, float cy, int align_corners, int normalize) {
    unsigned bx, by;
    xcd_band_block(bx, by);
    const int b = (int)by;
    const int pix = (int)(bx * blockDim.x + threadIdx.x);
    if (pix >= H * W) return;
    const int Y = pix / W, X = pix - Y * W;
    const float* p = params + (size_t)b * VIDC_WARP_PARAMS;
    const float px_min = p[27], py_min = p[28], kw = p[29], kh = p[30];
    float u, v;
    {
#pragma clang fp contract(off)
        float P0 = (p[0] * (float)X + p[1] * (float)Y) + p[2];
        float P1 = (p[3] * (float)X + p[4] * (float)Y) + p[5];
        float P2 = (p[6] * (float)X + p[7] * (float)Y) + p[8];
        u = kw * (P0 / P2 - px_min);
        v = kh * (P1 / P2 - py_min);
    }
    Taps t = make_taps(u, v, cx, cy, W, H, align_corners);
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * 3 * plane;
    float y0 = vidc::sample_nt(xb, t), y1 = vidc::sample_nt(xb + plane, t), y2 = vidc::sample_nt(xb + 2 * plane, t);      // (x: the head's output of this tick)
    // z = R^T y  (C_R_Cg.bmm(y), warping_2dof_alignment.py:253)
    float z0 = p[9] * y0 + p[12] * y1 + p[15] * y2;
    float z1 = p[10] * y0 + p[13] * y1 + p[16] * y2;
    float z2 = p[11] * y0 + p[14] * y1 + p[17] * y2;
    if (normalize) {   // F.normalize(dim=1): v / max(||v||_2, 1e-12)
        float n = fmaxf(sqrtf(z0 * z0 + z1 * z1 + z2 * z2), 1e-12f);
        z0 /= n; z1 /= n; z2 /= n;
    }
    float* zb = z + (size_t)b * 3 * plane + pix;
    zb[0] = z0; zb[plane] = z1; zb[2 * plane] = z2;
}

}  // namespace

extern "C" int vidc_warp2dof_params(const float* gravity, const float* aligned, int B, float fx, float fy, float cx, float cy,
                                    const float* K_inv, int W, int H, float* params, vidc_stream_t stream) {
    VIDC_REQUIRE(gravity && aligned && K_inv && params, VIDC_ERR_NULL, "vidc_warp2dof_params: null pointer");
    VIDC_REQUIRE(B > 0 && W > 1 && H > 1, VIDC_ERR_SHAPE, "vidc_warp2dof_params: bad shape B=%d W=%d H=%d", B, W, H);
    hipLaunchKernelGGL(warp_params_kernel, dim3(vidc::cdiv(B, 64)), dim3(64), 0, vidc::as_stream(stream), gravity, aligned, B,
                       fx, fy, cx, cy, K_inv, W, H, params);
    VIDC_CHECK_LAUNCH("warp_params_kernel");
    return VIDC_OK;
}

extern "C" int vidc_warp2dof_fwd(const float* x, const float* params, float* y, int B, int C, int H, int W, float cx, float cy,
                                 int align_corners, vidc_stream_t stream) {
    VIDC_REQUIRE(x && params && y, VIDC_ERR_NULL, "vidc_warp2dof_fwd: null pointer");
    VIDC_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, VIDC_ERR_SHAPE, "vidc_warp2dof_fwd: bad shape");
    hipLaunchKernelGGL(warp_fwd_kernel, dim3((vidc::cdiv(H * W, 256) + 7) / 8 * 8, B), dim3(256), 0, vidc::as_stream(stream), x, params, y, C,
                       H, W, cx, cy, align_corners);
    VIDC_CHECK_LAUNCH("warp_fwd_kernel");
    return VIDC_OK;
}

extern "C" int vidc_warp2dof_inv_rot_norm(const float* x, const float* params, float* z, int B, int H, int W, float cx,
                                          float cy, int align_corners, int normalize, vidc_stream_t stream) {
    VIDC_REQUIRE(x && params && z, VIDC_ERR_NULL, "vidc_warp2dof_inv_rot_norm: null pointer");
    VIDC_REQUIRE(B > 0 && H > 1 && W > 1, VIDC_ERR_SHAPE, "vidc_warp2dof_inv_rot_norm: bad shape");
    hipLaunchKernelGGL(warp_inv_rot_norm_kernel, dim3((vidc::cdiv(H * W, 256) + 7) / 8 * 8, B), dim3(256), 0, vidc::as_stream(stream), x,
                       params, z, H, W, cx, cy, align_corners, normalize);
    VIDC_CHECK_LAUNCH("warp_inv_rot_norm_kernel");
    return VIDC_OK;
}
