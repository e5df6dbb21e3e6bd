// Winograd F(m x m, 3x3), m = 2 or 4, around the MFMA GEMM kernel of conv_mfma.hip: the 3x3 / stride-1 / pad-1 Conv2d + BatchNorm2d + ReLU
// layers of the two decoders (networks/surface_normal.py:73-145, networks/depth_completion.py:75-147) are two thirds of the
// reference's conv FLOPs (one 768 -> 768 3x3 at 60x80 alone is 51 of 294 GFLOP).  In the Winograd domain such a layer is
// a^2 = (m + 2)^2 independent GEMMs  M[pos] = V[pos] (tiles x Cin)  *  U[pos] (Cin x Cout),  pos = (xi, nu),  with
//     V = B^T d B   (d: the a x a input patch of an m x m output tile, zero padded),
//     U = G g G^T   (g: the 3x3 filter; transformed once per checkpoint, in fp64),
//     Y = A^T M A   (the m x m outputs of the tile),
// i.e. 16 / 4 = 4 (m = 2) or 36 / 16 = 2.25 (m = 4) multiplications per output and input channel instead of 9.  The GEMMs run on
// the existing kernel as ONE grouped 1x1 "conv" with a^2 * G groups (vidc_conv2d_bn_act, identity epilogue); this file holds the
// three HBM-bound transforms.  Arithmetic is fp32 throughout (fp64 for U); measured on the oracle's CPU path the whole-frame depth
// changes by RMSE 1.0e-6 (m = 2) / 1.3e-6 (m = 4) against the direct form -- the same size as the fp32 summation-order differences between
// the direct HIP kernel and ATen (bar: 1e-3).
//
// Layouts (C = G * Cin resp. G * Cout channels of the NHWC tensor, group gg at channel gg * Cin):
//     V, M : [tile][gg][pos][Cin | Cout] fp32 rows of a^2 * C values (V optionally as the split-bf16 image of the same bytes), tile =
//            (b * th + ty) * tw + tx with th = ceil(H / m), tw = ceil(W / m): GEMM group gg * a^2 + pos reads channel slice
//            [(gg * a^2 + pos) * Cin, +Cin) of every row, so a restriction to a contiguous range of gg is a pointer offset.
//     U    : [gg][pos][Cout][Cin]
// One thread = one tile x 4 consecutive channels (16-byte accesses, lanes along channels).
#include "common.h"

namespace {

typedef float4 f4;
__device__ __forceinline__ f4 operator+(f4 a, f4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ f4 operator-(f4 a, f4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ f4 operator*(float s, f4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// B^T d for one column (input transform) and A^T m for one column (output transform); the standard matrices of Lavin & Gray
// ("Fast Algorithms for Convolutional Neural Networks", 2015) with interpolation points 0, +-1 (m = 2) and 0, +-1, +-2 (m = 4).
template <int M_> struct Wino;
template <> struct Wino<2> {
    static constexpr int A = 4;
    __device__ static __forceinline__ void bt(const f4 (&d)[4], f4 (&t)[4]) {
        t[0] = d[0] - d[2];
        t[1] = d[1] + d[2];
        t[2] = d[2] - d[1];
        t[3] = d[1] - d[3];
    }
    __device__ static __forceinline__ void at(const f4 (&m)[4], f4 (&o)[2]) {
        o[0] = (m[0] + m[1]) + m[2];
        o[1] = (m[1] - m[2]) - m[3];
    }
};
template <> struct Wino<4> {
    static constexpr int A = 6;
    __device__ static __forceinline__ void bt(const f4 (&d)[6], f4 (&t)[6]) {
        const f4 p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1];
        const f4 r = d[4] - d[2], s = 2.f * (d[3] - d[1]);
        t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
        t[1] = p + q;
        t[2] = p - q;
        t[3] = r + s;
        t[4] = r - s;
        t[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
    }
    __device__ static __forceinline__ void at(const f4 (&m)[6], f4 (&o)[4]) {
        const f4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        o[0] = (m[0] + s12) + s34;
        o[1] = d12 + 2.f * d34;
        o[2] = s12 + 4.f * s34;
        o[3] = (d12 + 8.f * d34) + m[5];
    }
};

// XCD k takes the k-th contiguous band of the (tile row, column block) space (see pointwise.hip xcd_band_block: neighbouring tile rows
// share a - m input rows, which then hit the same L2).  gridDim.x is a multiple of 8.
__device__ __forceinline__ void band_block(unsigned& bx, unsigned& by) {
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, per = (gx >> 3) * gridDim.y;
    const unsigned idx = (L & 7u) * per + (L >> 3);
    by = idx / gx;
    bx = idx - by * gx;
}

template <int M_>
__global__ void __launch_bounds__(256)
wino_in_kernel(const float* __restrict__ x, float* __restrict__ v, unsigned short* __restrict__ vsp, int H, int W, int C, int ldx, int Cin,
               int th, int tw, int ldv) {
    constexpr int A = Wino<M_>::A;
    const unsigned q = (unsigned)C / 4u;
    unsigned bx, by;                       // by = tile row (b, ty): wave-uniform
    band_block(bx, by);
    const unsigned i = bx * blockDim.x + threadIdx.x;
    if (i >= (unsigned)tw * q) return;
    const int tx = (int)(i / q);
    const int c = (int)(i - (unsigned)tx * q) * 4;
    const int b = (int)(by / (unsigned)th), ty = (int)(by - (unsigned)b * (unsigned)th);
    const int iy0 = ty * M_ - 1, ix0 = tx * M_ - 1;
    f4 d[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r) {
        const int iy = iy0 + r;
        const bool rok = (unsigned)iy < (unsigned)H;
#pragma unroll
        for (int s = 0; s < A; ++s) {
            const int ix = ix0 + s;
            d[r][s] = (rok && (unsigned)ix < (unsigned)W) ? *reinterpret_cast<const f4*>(&x[((size_t)(b * H + iy) * W + ix) * ldx + c])
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // B^T d (columns), then (.) B (rows)
    f4 t[A][A];
#pragma unroll
    for (int s = 0; s < A; ++s) {
        f4 col[A], out[A];
#pragma unroll
        for (int r = 0; r < A; ++r) col[r] = d[r][s];
        Wino<M_>::bt(col, out);
#pragma unroll
        for (int r = 0; r < A; ++r) t[r][s] = out[r];
    }
    const int gg = c / Cin, cc = c - gg * Cin;
    const size_t tile = ((size_t)by) * tw + tx;
    const int ch0 = gg * (A * A) * Cin + cc;
#pragma unroll
    for (int r = 0; r < A; ++r) {
        f4 out[A];
        Wino<M_>::bt(t[r], out);
#pragma unroll
        for (int s = 0; s < A; ++s) {
            const int ch = ch0 + (r * A + s) * Cin;
            if (vsp) vidc::store_split4(vsp, tile, ldv, ch, out[s]);
            else *reinterpret_cast<f4*>(&v[tile * ldv + ch]) = out[s];
        }
    }
}

template <int M_>
__global__ void __launch_bounds__(256)
wino_out_kernel(const float* __restrict__ mm, float* __restrict__ y, unsigned short* __restrict__ ysp, const float* __restrict__ scale1,
                const float* __restrict__ shift1, const float* __restrict__ scale2, const float* __restrict__ shift2, int Ho, int Wo, int C,
                int Cout, int ldy, int th, int tw, int flags, int ldm) {
    constexpr int A = Wino<M_>::A;
    const unsigned q = (unsigned)C / 4u;
    unsigned bx, by;
    band_block(bx, by);
    const unsigned i = bx * blockDim.x + threadIdx.x;
    if (i >= (unsigned)tw * q) return;
    const int tx = (int)(i / q);
    const int c = (int)(i - (unsigned)tx * q) * 4;
    const int b = (int)(by / (unsigned)th), ty = (int)(by - (unsigned)b * (unsigned)th);
    const int gg = c / Cout, cc = c - gg * Cout;
    const size_t tile = ((size_t)by) * tw + tx;
    const float* src = mm + tile * ldm + gg * (A * A) * Cout + cc;
    f4 m[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
        for (int s = 0; s < A; ++s) m[r][s] = *reinterpret_cast<const f4*>(&src[(r * A + s) * Cout]);
    // A^T m (columns), then (.) A (rows)
    f4 t[M_][A];
#pragma unroll
    for (int s = 0; s < A; ++s) {
        f4 col[A], out[M_];
#pragma unroll
        for (int r = 0; r < A; ++r) col[r] = m[r][s];
        Wino<M_>::at(col, out);
#pragma unroll
        for (int r = 0; r < M_; ++r) t[r][s] = out[r];
    }
    // the epilogue of conv_mfma.hip: acc * s1 + b1 -> relu -> [* s2 + b2 -> relu]; ReLU off = max with -inf
    const f4 s1 = *reinterpret_cast<const f4*>(&scale1[c]), b1 = *reinterpret_cast<const f4*>(&shift1[c]);
    const float lo1 = (flags & VIDC_RELU1) ? 0.f : -INFINITY, lo2 = (flags & VIDC_RELU2) ? 0.f : -INFINITY;
    const bool aff2 = flags & VIDC_AFFINE2;
    f4 s2 = make_float4(1.f, 1.f, 1.f, 1.f), b2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aff2) { s2 = *reinterpret_cast<const f4*>(&scale2[c]); b2 = *reinterpret_cast<const f4*>(&shift2[c]); }
    const bool st_f32 = !(flags & VIDC_NO_F32_OUT);
#pragma unroll
    for (int r = 0; r < M_; ++r) {
        f4 out[M_];
        Wino<M_>::at(t[r], out);
        const int oy = ty * M_ + r;
        if (oy >= Ho) continue;
#pragma unroll
        for (int s = 0; s < M_; ++s) {
            const int ox = tx * M_ + s;
            if (ox >= Wo) continue;
            f4 o = out[s];
            o.x = fmaxf(o.x * s1.x + b1.x, lo1); o.y = fmaxf(o.y * s1.y + b1.y, lo1);
            o.z = fmaxf(o.z * s1.z + b1.z, lo1); o.w = fmaxf(o.w * s1.w + b1.w, lo1);
            if (aff2) {
                o.x = fmaxf(o.x * s2.x + b2.x, lo2); o.y = fmaxf(o.y * s2.y + b2.y, lo2);
                o.z = fmaxf(o.z * s2.z + b2.z, lo2); o.w = fmaxf(o.w * s2.w + b2.w, lo2);
            }
            const size_t row = (size_t)(b * Ho + oy) * Wo + ox;
            if (st_f32) *reinterpret_cast<f4*>(&y[row * ldy + c]) = o;
            if (ysp) vidc::store_split4(ysp, row, ldy, c, o);
        }
    }
}

// U = G g G^T in fp64, rounded once to fp32.  One thread per (co, ci).
template <int M_>
__global__ void __launch_bounds__(256)
wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
    constexpr int A = M_ + 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    double G[A][3];
    if constexpr (M_ == 2) {
        const double g_[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
        for (int i = 0; i < A; ++i) for (int j = 0; j < 3; ++j) G[i][j] = g_[i][j];
    } else {
        const double g_[6][3] = {{1. / 4, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
        for (int i = 0; i < A; ++i) for (int j = 0; j < 3; ++j) G[i][j] = g_[i][j];
    }
    double g[3][3];
    for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = (double)w[idx * 9 + k];
    double t[A][3];
    for (int i = 0; i < A; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
    for (int i = 0; i < A; ++i)
        for (int j = 0; j < A; ++j) {
            const double v = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
            u[(size_t)(i * A + j) * Cout * Cin + idx] = (float)v;
        }
}

}  // namespace

extern "C" int vidc_winograd_tiles(int H, int W, int m, int* th, int* tw) {
    VIDC_REQUIRE(th && tw, VIDC_ERR_NULL, "vidc_winograd_tiles: null pointer");
    VIDC_REQUIRE((m == 2 || m == 4) && H > 0 && W > 0, VIDC_ERR_SHAPE, "vidc_winograd_tiles: m must be 2 or 4");
    *th = (H + m - 1) / m;
    *tw = (W + m - 1) / m;
    return VIDC_OK;
}

extern "C" int vidc_winograd_weight_transform(const float* w_oihw, float* u, int Cout, int Cin, int m, vidc_stream_t stream) {
    VIDC_REQUIRE(w_oihw && u, VIDC_ERR_NULL, "vidc_winograd_weight_transform: null pointer");
    VIDC_REQUIRE((m == 2 || m == 4) && Cout > 0 && Cin > 0, VIDC_ERR_SHAPE, "vidc_winograd_weight_transform: m must be 2 or 4");
    const unsigned blocks = (unsigned)(((long long)Cout * Cin + 255) / 256);
    if (m == 2) hipLaunchKernelGGL(wino_weight_kernel<2>, dim3(blocks), dim3(256), 0, vidc::as_stream(stream), w_oihw, u, Cout, Cin);
    else hipLaunchKernelGGL(wino_weight_kernel<4>, dim3(blocks), dim3(256), 0, vidc::as_stream(stream), w_oihw, u, Cout, Cin);
    VIDC_CHECK_LAUNCH("wino_weight_kernel");
    return VIDC_OK;
}

extern "C" int vidc_winograd_input_transform(const float* x, void* v, int B, int H, int W, int C, int ldx, int Cin, int m, int split,
                                             int ldv, vidc_stream_t stream) {
    VIDC_REQUIRE(x && v, VIDC_ERR_NULL, "vidc_winograd_input_transform: null pointer");
    VIDC_REQUIRE(m == 2 || m == 4, VIDC_ERR_SHAPE, "vidc_winograd_input_transform: m must be 2 or 4");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Cin > 0 && C % Cin == 0 && Cin % 4 == 0 && ldx % 4 == 0 && ldx >= C, VIDC_ERR_SHAPE,
                 "vidc_winograd_input_transform: bad shape (C = G * Cin, Cin and ldx multiples of 4)");
    VIDC_REQUIRE(!split || Cin % 32 == 0, VIDC_ERR_SHAPE, "vidc_winograd_input_transform: split output needs Cin % 32 == 0");
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    if (ldv == 0) ldv = (m + 2) * (m + 2) * C;
    VIDC_REQUIRE(ldv >= (m + 2) * (m + 2) * C && ldv % 4 == 0 && (!split || ldv % 32 == 0), VIDC_ERR_SHAPE, "vidc_winograd_input_transform: bad row stride ldv = %d", ldv);
    VIDC_REQUIRE((long long)B * th <= 65535, VIDC_ERR_SHAPE, "vidc_winograd_input_transform: B * tile rows = %lld exceed the grid", (long long)B * th);
    const dim3 grid((unsigned)(vidc::cdiv(tw * (C / 4), 256) + 7) / 8 * 8, (unsigned)(B * th));
    float* vf = split ? nullptr : reinterpret_cast<float*>(v);
    unsigned short* vs = split ? reinterpret_cast<unsigned short*>(v) : nullptr;
    if (m == 2) hipLaunchKernelGGL(wino_in_kernel<2>, grid, dim3(256), 0, vidc::as_stream(stream), x, vf, vs, H, W, C, ldx, Cin, th, tw, ldv);
    else hipLaunchKernelGGL(wino_in_kernel<4>, grid, dim3(256), 0, vidc::as_stream(stream), x, vf, vs, H, W, C, ldx, Cin, th, tw, ldv);
    VIDC_CHECK_LAUNCH("wino_in_kernel");
    return VIDC_OK;
}

extern "C" int vidc_winograd_output_transform(const float* mm, float* y, void* y_split, const float* scale1, const float* shift1,
                                              const float* scale2, const float* shift2, int B, int Ho, int Wo, int C, int Cout, int ldy,
                                              int m, int flags, int ldm, vidc_stream_t stream) {
    VIDC_REQUIRE(mm && scale1 && shift1 && (y || y_split), VIDC_ERR_NULL, "vidc_winograd_output_transform: null pointer");
    VIDC_REQUIRE(m == 2 || m == 4, VIDC_ERR_SHAPE, "vidc_winograd_output_transform: m must be 2 or 4");
    VIDC_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && Cout > 0 && C % Cout == 0 && Cout % 4 == 0 && ldy % 4 == 0 && ldy >= C, VIDC_ERR_SHAPE,
                 "vidc_winograd_output_transform: bad shape (C = G * Cout, Cout and ldy multiples of 4)");
    VIDC_REQUIRE(!(flags & ~(VIDC_RELU1 | VIDC_AFFINE2 | VIDC_RELU2 | VIDC_SPLIT_OUT | VIDC_NO_F32_OUT)), VIDC_ERR_SHAPE,
                 "vidc_winograd_output_transform: unsupported flags 0x%x", flags);
    VIDC_REQUIRE(!(flags & VIDC_AFFINE2) || (scale2 && shift2), VIDC_ERR_NULL, "vidc_winograd_output_transform: AFFINE2 without scale2 / shift2");
    VIDC_REQUIRE(!!(flags & VIDC_SPLIT_OUT) == (y_split != nullptr), VIDC_ERR_SHAPE, "vidc_winograd_output_transform: SPLIT_OUT and y_split must come together");
    VIDC_REQUIRE(!(flags & VIDC_NO_F32_OUT) || y_split, VIDC_ERR_SHAPE, "vidc_winograd_output_transform: NO_F32_OUT without y_split writes nothing");
    VIDC_REQUIRE((flags & VIDC_NO_F32_OUT) || y, VIDC_ERR_NULL, "vidc_winograd_output_transform: y is NULL");
    VIDC_REQUIRE(!y_split || ldy % 32 == 0, VIDC_ERR_SHAPE, "vidc_winograd_output_transform: split output needs ldy % 32 == 0");
    const int th = (Ho + m - 1) / m, tw = (Wo + m - 1) / m;
    if (ldm == 0) ldm = (m + 2) * (m + 2) * C;
    VIDC_REQUIRE(ldm >= (m + 2) * (m + 2) * C && ldm % 4 == 0, VIDC_ERR_SHAPE, "vidc_winograd_output_transform: bad row stride ldm = %d", ldm);
    VIDC_REQUIRE((long long)B * th <= 65535, VIDC_ERR_SHAPE, "vidc_winograd_output_transform: B * tile rows = %lld exceed the grid", (long long)B * th);
    const dim3 grid((unsigned)(vidc::cdiv(tw * (C / 4), 256) + 7) / 8 * 8, (unsigned)(B * th));
    unsigned short* ys = reinterpret_cast<unsigned short*>(y_split);
    if (m == 2) hipLaunchKernelGGL(wino_out_kernel<2>, grid, dim3(256), 0, vidc::as_stream(stream), mm, y, ys, scale1, shift1, scale2, shift2, Ho, Wo, C, Cout, ldy, th, tw, flags, ldm);
    else hipLaunchKernelGGL(wino_out_kernel<4>, grid, dim3(256), 0, vidc::as_stream(stream), mm, y, ys, scale1, shift1, scale2, shift2, Ho, Wo, C, Cout, ldy, th, tw, flags, ldm);
    VIDC_CHECK_LAUNCH("wino_out_kernel");
    return VIDC_OK;
}
