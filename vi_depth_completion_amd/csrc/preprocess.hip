// Device-side frame pre-processing: what the reference's DemoDataset.__getitem__ does on the host per frame
// (dataset.py:461-510) -- PIL bilinear resize + ToTensor, and rasterisation of the VI-SLAM sparse points -- so that a stream of
// camera frames does not go through a Python/PIL DataLoader at several hundred frames per second.
//
// Resize: bit-identical to Pillow's Image.resize(size, Image.BILINEAR) on 8-bit images (src/libImaging/Resample.c: triangle filter
// with support 1.0 x down-scale factor, 22-bit fixed-point coefficients, horizontal pass, 8-bit intermediate, vertical pass), fused
// with transforms.ToTensor (HWC uint8 -> CHW float / 255).  Both passes run in ONE kernel: every output element recomputes the
// (<= ksize_y) horizontally filtered, clipped 8-bit values it needs -- HBM-bound, one read of the source and one write of the result.
#include "common.h"
#include <cmath>
#include <cstdint>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;        // Resample.c

__device__ inline int clip8(int v) {              // Resample.c clip8(): (in >> PRECISION_BITS) clamped to 0..255
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One workgroup per output row: the <= ksy source rows it needs are staged in LDS with 16-byte loads (each source row is read
// from HBM/L2 by the ~2 output rows that use it), then one thread per (channel, output column) runs both passes out of LDS;
// consecutive lanes write consecutive columns of one channel plane.
__global__ void __launch_bounds__(256)
resize_bilinear_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int H, int W, int C, int Ho, int Wo,
                          const int32_t* __restrict__ bx, const int32_t* __restrict__ kx, int ksx,
                          const int32_t* __restrict__ by, const int32_t* __restrict__ ky, int ksy, int row_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];     // [ksy][row_pitch]
    const int yo = blockIdx.x, b = blockIdx.y;
    const int ymin = by[2 * yo], ny = by[2 * yo + 1];
    const int row_bytes = W * C;
    const uint8_t* sb = src + ((size_t)b * H + ymin) * row_bytes;
    if ((row_bytes & 15) == 0 && ((size_t)sb & 15) == 0) {
        const int n16 = row_bytes >> 4;
        for (int i = threadIdx.x; i < ny * n16; i += blockDim.x) {
            const int r = i / n16, q = i - r * n16;
            *reinterpret_cast<uint4*>(rows + r * row_pitch + q * 16) = *reinterpret_cast<const uint4*>(sb + (size_t)r * row_bytes + q * 16);
        }
    } else {
        for (int i = threadIdx.x; i < ny * row_bytes; i += blockDim.x) {
            const int r = i / row_bytes, q = i - r * row_bytes;
            rows[r * row_pitch + q] = sb[(size_t)r * row_bytes + q];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * Wo; idx += blockDim.x) {
        const int c = idx / Wo, xo = idx - c * Wo;
        const int xmin = bx[2 * xo], nx = bx[2 * xo + 1];
        const int32_t* kxo = kx + xo * ksx;
        int sv = 1 << (PRECISION_BITS - 1);
        for (int r = 0; r < ny; ++r) {
            const uint8_t* row = rows + r * row_pitch + xmin * C + c;
            int sh = 1 << (PRECISION_BITS - 1);
            for (int x = 0; x < nx; ++x) sh += (int)row[x * C] * kxo[x];
            sv += clip8(sh) * ky[yo * ksy + r];           // the 8-bit intermediate image of the two-pass algorithm
        }
        dst[(((size_t)b * C + c) * Ho + yo) * Wo + xo] = (float)clip8(sv) / 255.0f;      // ToTensor: byte / 255 in fp32 (IEEE division)
    }
}

// One workgroup per image: zero the map, then ONE lane replays the reference's sequential loop (a later track overwrites an
// earlier one on the same pixel, dataset.py:497-510).  A few hundred tracks per frame: bounded, tiny.
__global__ void __launch_bounds__(256)
rasterize_sparse_kernel(const double* __restrict__ tracks, const int32_t* __restrict__ offsets, double fx, double fy, double cx,
                        double cy, float* __restrict__ depth, int H, int W) {
    const int b = blockIdx.x;
    float* d = depth + (size_t)b * H * W;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) d[i] = 0.f;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int i = offsets[b]; i < offsets[b + 1]; ++i) {
        const double X = tracks[4 * i + 1], Y = tracks[4 * i + 2], Z = tracks[4 * i + 3];
        const double u = X / Z, v = Y / Z;
        const double px = __dadd_rn(__dmul_rn(fx, u), cx), py = __dadd_rn(__dmul_rn(fy, v), cy);   // numpy: multiply, then add (no FMA)
        if (!(px > -2147483000.0 && px < 2147483000.0 && py > -2147483000.0 && py < 2147483000.0)) continue;   // NaN / huge
        const int col = (int)px, row = (int)py;                                                    // int(): truncation toward zero
        if (row >= 0 && row < H && col >= 0 && col < W) d[row * W + col] = (float)Z;
    }
}

// Ground-truth depth of the training / evaluation streams (dataset.py:283-286: 16-bit millimetre PNG -> convert('F') -> Image.resize(...,
// NEAREST) -> / 1000.0): one gather per output pixel through the two index tables of vidc_nearest_table, IEEE fp32 division.
__global__ void __launch_bounds__(256)
nearest_u16_depth_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int H, int W, int Ho, int Wo,
                         const int32_t* __restrict__ xtab, const int32_t* __restrict__ ytab, float divisor) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Ho * Wo) return;
    const int yo = idx / Wo, xo = idx - yo * Wo;
    const int ys = ytab[yo], xs = xtab[xo];
    // (a table entry outside the image is Pillow's "fill" case: the output pixel stays 0; it cannot occur for a whole-image box)
    const float v = (ys >= 0 && ys < H && xs >= 0 && xs < W) ? (float)src[((size_t)b * H + ys) * W + xs] : 0.f;
    dst[(size_t)b * Ho * Wo + idx] = __fdiv_rn(v, divisor);
}

}  // namespace

// Host function (no GPU): the source index of every output coordinate of Pillow's Image.resize(..., NEAREST) along one axis
// (src/libImaging/Geometry.c ImagingScaleAffine: xo = a0 * 0.5, then xo += a0 per output pixel IN DOUBLE -- the accumulated sum, not
// (x + 0.5) * a0 --, index = xo < 0 ? -1 : (int)xo).
extern "C" int vidc_nearest_table(int in_size, int out_size, int32_t* table) {
    VIDC_REQUIRE(in_size > 0 && out_size > 0 && table, VIDC_ERR_SHAPE, "vidc_nearest_table: bad arguments");
    const double a0 = (double)in_size / (double)out_size;
    double xo = a0 * 0.5;
    for (int x = 0; x < out_size; ++x) {
        table[x] = xo < 0.0 ? -1 : (int)xo;
        xo += a0;
    }
    return VIDC_OK;
}

extern "C" int vidc_resize_nearest_u16_depth(const uint16_t* src, float* dst, int B, int H, int W, int Ho, int Wo, const int32_t* xtab,
                                             const int32_t* ytab, float divisor, vidc_stream_t stream) {
    VIDC_REQUIRE(src && dst && xtab && ytab, VIDC_ERR_NULL, "vidc_resize_nearest_u16_depth: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && divisor != 0.f, VIDC_ERR_SHAPE, "vidc_resize_nearest_u16_depth: bad shape");
    dim3 grid((Ho * Wo + 255) / 256, B, 1);
    hipLaunchKernelGGL(nearest_u16_depth_kernel, grid, dim3(256), 0, vidc::as_stream(stream), src, dst, H, W, Ho, Wo, xtab, ytab, divisor);
    VIDC_CHECK_LAUNCH("nearest_u16_depth_kernel");
    return VIDC_OK;
}

// Host function (no GPU): Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the full input range.
extern "C" int vidc_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int coeffs_capacity, int* ksize_out) {
    VIDC_REQUIRE(in_size > 0 && out_size > 0 && ksize_out, VIDC_ERR_SHAPE, "vidc_resize_coeffs: bad sizes");
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;               // bilinear: support 1.0
    const int ksize = (int)ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!bounds || !coeffs) return VIDC_OK;                 // size query
    VIDC_REQUIRE(coeffs_capacity >= out_size * ksize, VIDC_ERR_SHAPE, "vidc_resize_coeffs: coeffs buffer too small (%d < %d)", coeffs_capacity,
                 out_size * ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double kk[64];
        VIDC_REQUIRE(ksize <= 64, VIDC_ERR_SHAPE, "vidc_resize_coeffs: down-scale factor too large");
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            kk[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) kk[x] /= ww;
        for (int x = xmax; x < ksize; ++x) kk[x] = 0.0;
        for (int x = 0; x < ksize; ++x)
            coeffs[xx * ksize + x] = kk[x] < 0 ? (int)(-0.5 + kk[x] * (1 << PRECISION_BITS)) : (int)(0.5 + kk[x] * (1 << PRECISION_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return VIDC_OK;
}

extern "C" int vidc_resize_bilinear_u8_to_chw(const uint8_t* src_hwc, float* dst_chw, int B, int H, int W, int C, int Ho, int Wo,
                                              const int32_t* bounds_x, const int32_t* coeffs_x, int ksize_x, const int32_t* bounds_y,
                                              const int32_t* coeffs_y, int ksize_y, vidc_stream_t stream) {
    VIDC_REQUIRE(src_hwc && dst_chw && bounds_x && coeffs_x && bounds_y && coeffs_y, VIDC_ERR_NULL, "vidc_resize_bilinear_u8_to_chw: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C <= 4 && Ho > 0 && Wo > 0 && ksize_x > 0 && ksize_y > 0, VIDC_ERR_SHAPE,
                 "vidc_resize_bilinear_u8_to_chw: bad shape");
    const int row_pitch = (W * C + 15) / 16 * 16;
    const size_t lds = (size_t)ksize_y * row_pitch;
    VIDC_REQUIRE(lds <= 64 * 1024, VIDC_ERR_SHAPE, "vidc_resize_bilinear_u8_to_chw: %d source rows of %d bytes do not fit in LDS", ksize_y, row_pitch);
    dim3 grid(Ho, B, 1);
    hipLaunchKernelGGL(resize_bilinear_u8_kernel, grid, dim3(256), lds, vidc::as_stream(stream), src_hwc, dst_chw, H, W, C, Ho, Wo, bounds_x,
                       coeffs_x, ksize_x, bounds_y, coeffs_y, ksize_y, row_pitch);
    VIDC_CHECK_LAUNCH("resize_bilinear_u8_kernel");
    return VIDC_OK;
}

extern "C" int vidc_rasterize_sparse_depth(const double* tracks, const int32_t* offsets, int B, double fx, double fy, double cx, double cy,
                                           float* depth, int H, int W, vidc_stream_t stream) {
    VIDC_REQUIRE(offsets && depth, VIDC_ERR_NULL, "vidc_rasterize_sparse_depth: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0, VIDC_ERR_SHAPE, "vidc_rasterize_sparse_depth: bad shape");
    hipLaunchKernelGGL(rasterize_sparse_kernel, dim3(B), dim3(256), 0, vidc::as_stream(stream), tracks, offsets, fx, fy, cx, cy, depth, H, W);
    VIDC_CHECK_LAUNCH("rasterize_sparse_kernel");
    return VIDC_OK;
}
