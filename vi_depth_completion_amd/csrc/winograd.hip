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
// One thread = one tile x 4 consecutive channels (16-byte accesses, lanes along channels), x 1 channel on the small maps (vecn below).
#include "common.h"

// No mul + add contraction anywhere in this file: the transforms are instantiated per tile size and per channels-per-thread, the
// restricted-group variants of a frame program (engine.Program.group_variant) pick another instantiation than the full launch for the
// same layer, and hipcc contracts `a - 4 * b` into an fma in one instantiation and not in the other -- a frame's bits then depended on
// which tick of a lane computed it (tests/test_frames_per_launch.py).  Every product and sum below rounds once, like the numpy oracle.
#pragma clang fp contract(off)

namespace {

// N consecutive channels per thread: 4 (16-byte accesses) on maps large enough to fill the chip that way, 1 on the small maps (a 16x20
// map of 1024 channels is 80 tiles x 256 float4 columns = 80 workgroups for 256 CUs, each thread a serial chain of 36 loads + 36 stores:
// launch / latency-bound at ~10 us; one channel per thread is four times the threads with a quarter of the chain, lanes still along channels)
template <int N> struct vecn {
    float v[N];
};
template <int N> __device__ __forceinline__ vecn<N> operator+(const vecn<N>& a, const vecn<N>& b) {
    vecn<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
template <int N> __device__ __forceinline__ vecn<N> operator-(const vecn<N>& a, const vecn<N>& b) {
    vecn<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = a.v[i] - b.v[i];
    return r;
}
template <int N> __device__ __forceinline__ vecn<N> operator*(float s, const vecn<N>& a) {
    vecn<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = s * a.v[i];
    return r;
}
template <int N> __device__ __forceinline__ vecn<N> vzero() {
    vecn<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = 0.f;
    return r;
}
template <int N> __device__ __forceinline__ vecn<N> vload(const float* p) {
    vecn<N> r;
    if constexpr (N == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = p[i];
    }
    return r;
}
template <int N> __device__ __forceinline__ void vstore(float* p, const vecn<N>& a) {
    if constexpr (N == 4) *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    else {
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = a.v[i];
    }
}
template <int N> __device__ __forceinline__ void vstore_split(unsigned short* img, size_t row, int ld, int c, const vecn<N>& a) {
    if constexpr (N == 4) vidc::store_split4(img, row, ld, c, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
    else {
#pragma unroll
        for (int i = 0; i < N; ++i) vidc::store_split(img, row, ld, c + i, a.v[i]);
    }
}

// B^T d for one column (input transform) and A^T m for one column (output transform); the standard matrices of Lavin & Gray
// ("Fast Algorithms for Convolutional Neural Networks", 2015) with interpolation points 0, +-1 (m = 2) and 0, +-1, +-2 (m = 4).
template <int M_> struct Wino;
template <> struct Wino<2> {
    static constexpr int A = 4;
    template <typename V> __device__ static __forceinline__ void bt(const V (&d)[4], V (&t)[4]) {
        t[0] = d[0] - d[2];
        t[1] = d[1] + d[2];
        t[2] = d[2] - d[1];
        t[3] = d[1] - d[3];
    }
    template <typename V> __device__ static __forceinline__ void at(const V (&m)[4], V (&o)[2]) {
        o[0] = (m[0] + m[1]) + m[2];
        o[1] = (m[1] - m[2]) - m[3];
    }
};
template <> struct Wino<4> {
    static constexpr int A = 6;
    template <typename V> __device__ static __forceinline__ void bt(const V (&d)[6], V (&t)[6]) {
        const V p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1];
        const V r = d[4] - d[2], s = 2.f * (d[3] - d[1]);
        t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
        t[1] = p + q;
        t[2] = p - q;
        t[3] = r + s;
        t[4] = r - s;
        t[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
    }
    template <typename V> __device__ static __forceinline__ void at(const V (&m)[6], V (&o)[4]) {
        const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
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

template <int M_, int N>
__global__ void __launch_bounds__(256)
wino_in_kernel(const float* __restrict__ x, float* __restrict__ v, unsigned short* __restrict__ vsp, int H, int W, int C, int ldx, int Cin,
               int th, int tw, int ldv) {
    constexpr int A = Wino<M_>::A;
    typedef vecn<N> V;
    const unsigned q = (unsigned)C / (unsigned)N;
    unsigned bx, by;                       // by = tile row (b, ty): wave-uniform
    band_block(bx, by);
    const unsigned i = bx * blockDim.x + threadIdx.x;
    if (i >= (unsigned)tw * q) return;
    const int tx = (int)(i / q);
    const int c = (int)(i - (unsigned)tx * q) * N;
    const int b = (int)(by / (unsigned)th), ty = (int)(by - (unsigned)b * (unsigned)th);
    const int iy0 = ty * M_ - 1, ix0 = tx * M_ - 1;
    V d[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r) {
        const int iy = iy0 + r;
        const bool rok = (unsigned)iy < (unsigned)H;
#pragma unroll
        for (int s = 0; s < A; ++s) {
            const int ix = ix0 + s;
            d[r][s] = (rok && (unsigned)ix < (unsigned)W) ? vload<N>(&x[((size_t)(b * H + iy) * W + ix) * ldx + c]) : vzero<N>();
        }
    }
    // B^T d (columns), then (.) B (rows)
    V t[A][A];
#pragma unroll
    for (int s = 0; s < A; ++s) {
        V col[A], out[A];
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
        V out[A];
        Wino<M_>::bt(t[r], out);
#pragma unroll
        for (int s = 0; s < A; ++s) {
            const int ch = ch0 + (r * A + s) * Cin;
            if (vsp) vstore_split<N>(vsp, tile, ldv, ch, out[s]);
            else vstore<N>(&v[tile * ldv + ch], out[s]);
        }
    }
}

template <int M_, int N>
__global__ void __launch_bounds__(256)
wino_out_kernel(const float* __restrict__ mm, float* __restrict__ y, unsigned short* __restrict__ ysp, const float* __restrict__ scale1,
                const float* __restrict__ shift1, const float* __restrict__ scale2, const float* __restrict__ shift2, int Ho, int Wo, int C,
                int Cout, int ldy, int th, int tw, int flags, int ldm) {
    constexpr int A = Wino<M_>::A;
    typedef vecn<N> V;
    const unsigned q = (unsigned)C / (unsigned)N;
    unsigned bx, by;
    band_block(bx, by);
    const unsigned i = bx * blockDim.x + threadIdx.x;
    if (i >= (unsigned)tw * q) return;
    const int tx = (int)(i / q);
    const int c = (int)(i - (unsigned)tx * q) * N;
    const int b = (int)(by / (unsigned)th), ty = (int)(by - (unsigned)b * (unsigned)th);
    const int gg = c / Cout, cc = c - gg * Cout;
    const size_t tile = ((size_t)by) * tw + tx;
    const float* src = mm + tile * ldm + gg * (A * A) * Cout + cc;
    V m[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
        for (int s = 0; s < A; ++s) m[r][s] = vload<N>(&src[(r * A + s) * Cout]);
    // A^T m (columns), then (.) A (rows)
    V t[M_][A];
#pragma unroll
    for (int s = 0; s < A; ++s) {
        V col[A], out[M_];
#pragma unroll
        for (int r = 0; r < A; ++r) col[r] = m[r][s];
        Wino<M_>::at(col, out);
#pragma unroll
        for (int r = 0; r < M_; ++r) t[r][s] = out[r];
    }
    // the epilogue of conv_mfma.hip: acc * s1 + b1 -> relu -> [* s2 + b2 -> relu]; ReLU off = max with -inf
    const V s1 = vload<N>(&scale1[c]), b1 = vload<N>(&shift1[c]);
    const float lo1 = (flags & VIDC_RELU1) ? 0.f : -INFINITY, lo2 = (flags & VIDC_RELU2) ? 0.f : -INFINITY;
    const bool aff2 = flags & VIDC_AFFINE2;
    V s2 = vzero<N>(), b2 = vzero<N>();
    if (aff2) { s2 = vload<N>(&scale2[c]); b2 = vload<N>(&shift2[c]); }
    const bool st_f32 = !(flags & VIDC_NO_F32_OUT);
#pragma unroll
    for (int r = 0; r < M_; ++r) {
        V out[M_];
        Wino<M_>::at(t[r], out);
        const int oy = ty * M_ + r;
        if (oy >= Ho) continue;
#pragma unroll
        for (int s = 0; s < M_; ++s) {
            const int ox = tx * M_ + s;
            if (ox >= Wo) continue;
            V o = out[s];
#pragma unroll
            for (int k = 0; k < N; ++k) o.v[k] = fmaxf(o.v[k] * s1.v[k] + b1.v[k], lo1);
            if (aff2) {
#pragma unroll
                for (int k = 0; k < N; ++k) o.v[k] = fmaxf(o.v[k] * s2.v[k] + b2.v[k], lo2);
            }
            const size_t row = (size_t)(b * Ho + oy) * Wo + ox;
            if (st_f32) vstore<N>(&y[row * ldy + c], o);
            if (ysp) vstore_split<N>(ysp, row, ldy, c, o);
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
    // four channels per thread where that still fills the chip, one on the small maps (see vecn)
    const bool wide = (long long)B * th * tw * (C / 4) >= 131072;
    const int per = wide ? 4 : 1;
    const dim3 grid((unsigned)(vidc::cdiv(tw * (C / per), 256) + 7) / 8 * 8, (unsigned)(B * th));
    float* vf = split ? nullptr : reinterpret_cast<float*>(v);
    unsigned short* vs = split ? reinterpret_cast<unsigned short*>(v) : nullptr;
    hipStream_t st = vidc::as_stream(stream);
    if (m == 2 && wide) hipLaunchKernelGGL((wino_in_kernel<2, 4>), grid, dim3(256), 0, st, x, vf, vs, H, W, C, ldx, Cin, th, tw, ldv);
    else if (m == 2) hipLaunchKernelGGL((wino_in_kernel<2, 1>), grid, dim3(256), 0, st, x, vf, vs, H, W, C, ldx, Cin, th, tw, ldv);
    else if (wide) hipLaunchKernelGGL((wino_in_kernel<4, 4>), grid, dim3(256), 0, st, x, vf, vs, H, W, C, ldx, Cin, th, tw, ldv);
    else hipLaunchKernelGGL((wino_in_kernel<4, 1>), grid, dim3(256), 0, st, x, vf, vs, H, W, C, ldx, Cin, th, tw, ldv);
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
    const bool wide = (long long)B * th * tw * (C / 4) >= 131072;
    const int per = wide ? 4 : 1;
    const dim3 grid((unsigned)(vidc::cdiv(tw * (C / per), 256) + 7) / 8 * 8, (unsigned)(B * th));
    unsigned short* ys = reinterpret_cast<unsigned short*>(y_split);
    hipStream_t st = vidc::as_stream(stream);
    if (m == 2 && wide) hipLaunchKernelGGL((wino_out_kernel<2, 4>), grid, dim3(256), 0, st, mm, y, ys, scale1, shift1, scale2, shift2, Ho, Wo, C, Cout, ldy, th, tw, flags, ldm);
    else if (m == 2) hipLaunchKernelGGL((wino_out_kernel<2, 1>), grid, dim3(256), 0, st, mm, y, ys, scale1, shift1, scale2, shift2, Ho, Wo, C, Cout, ldy, th, tw, flags, ldm);
    else if (wide) hipLaunchKernelGGL((wino_out_kernel<4, 4>), grid, dim3(256), 0, st, mm, y, ys, scale1, shift1, scale2, shift2, Ho, Wo, C, Cout, ldy, th, tw, flags, ldm);
    else hipLaunchKernelGGL((wino_out_kernel<4, 1>), grid, dim3(256), 0, st, mm, y, ys, scale1, shift1, scale2, shift2, Ho, Wo, C, Cout, ldy, th, tw, flags, ldm);
    VIDC_CHECK_LAUNCH("wino_out_kernel");
    return VIDC_OK;
}
