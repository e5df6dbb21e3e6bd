// Evaluation statistics and output conversion on the device.
//
// The reference copies two full error maps per batch to the host (`_network_evaluate`, network_run.py:198-225), concatenates them over
// the whole test set and reduces at the end (`evaluate`, network_run.py:387-403).  Here every batch is reduced on the device to 8
// numbers -- the sufficient statistics of exactly those final figures -- with a fixed two-level order (bit-reproducible), and only they
// travel: n valid (gt > 0), sum |gt - pred|, sum (gt - pred)^2, and the counts of max(gt/pred, pred/gt) below 1.05, 1.10, 1.25,
// 1.25^2, 1.25^3 (`GetDepthPrintableRatios`, network_run.py:72-82, and the DEPTH ERROR STATS line).
#include "common.h"
#include <cstdint>

namespace {

constexpr int MT = 256;                 // threads per workgroup
constexpr int MSTATS = 8;

__device__ inline double wg_sum(double v, double* red) {          // fixed order: lane tree, then waves 0..3
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// partial[blockIdx.x][8]; thread t of block b owns elements b*MT*PER + t + k*MT (coalesced), accumulated in fp64
__global__ void __launch_bounds__(MT)
depth_metrics_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long long n, int per, double* __restrict__ partial) {
    __shared__ double red[4];
    const float thr[5] = {1.05f, 1.10f, 1.25f, 1.25f * 1.25f, 1.25f * 1.25f * 1.25f};
    double s[MSTATS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long base = (long long)blockIdx.x * MT * per;
    for (int k = 0; k < per; ++k) {
        const long long i = base + (long long)k * MT + threadIdx.x;
        if (i >= n) break;
        const float g = gt[i], p = pred[i];
        if (!(g > 0.f)) continue;                        // depth_mask = depths_gt > 0
        const float e = fabsf(g - p);                    // fp32 like the reference's tensors
        const float r = fmaxf(g / p, p / g);             // torch.max(gt / pred, pred / gt); NaN compares false below, like numpy
        s[0] += 1.0;
        s[1] += (double)e;
        s[2] += (double)e * (double)e;
#pragma unroll
        for (int t = 0; t < 5; ++t) s[3 + t] += (r < thr[t]) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int q = 0; q < MSTATS; ++q) {
        const double v = wg_sum(s[q], red);
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * MSTATS + q] = v;
    }
}

// stats[8] (+)= sum over the partials in index order (one workgroup; thread t owns partials t, t+MT, ...)
__global__ void __launch_bounds__(MT)
depth_metrics_final_kernel(const double* __restrict__ partial, int n_partial, double* __restrict__ stats, int accumulate) {
    __shared__ double red[4];
#pragma unroll
    for (int q = 0; q < MSTATS; ++q) {
        double v = 0.0;
        for (int i = threadIdx.x; i < n_partial; i += MT) v += partial[(size_t)i * MSTATS + q];
        v = wg_sum(v, red);
        if (threadIdx.x == 0) stats[q] = (accumulate ? stats[q] : 0.0) + v;
    }
}

// SaveDepthsToImage (network_run.py:42-50): (depths * 1000).astype(np.uint32) -- fp32 product, truncation toward zero
__global__ void __launch_bounds__(MT) depth_to_mm_kernel(const float* __restrict__ depth, uint32_t* __restrict__ mm, long long n) {
    const long long i = (long long)blockIdx.x * MT + threadIdx.x;
    if (i >= n) return;
    const float v = depth[i] * 1000.0f;
    mm[i] = v >= 4294967296.0f ? 0xFFFFFFFFu : (v > 0.f ? (uint32_t)v : 0u);      // the path's depths are >= 0 (final ReLU)
}

// ---- normal-error statistics (network_run.py:204-214 + the NORMAL ERROR STATS line of evaluate(), :389-397) -------------------------
// Per valid pixel (mask > 0): angle = acos(clamp(<normalize(pred), normalize(gt)>, -1, 1)) / pi * 180 in fp32, like the reference's
// tensors.  The 8 sufficient statistics of the logged figures except the median -- n, sum e, sum e^2, counts below 5 / 7.5 / 11.25 /
// 22.5 / 30 degrees -- are reduced exactly like the depth ones; the angle of every pixel is also written out (invalid pixels get
// the all-ones bit pattern) so that the EXACT median can be selected later by two radix-histogram passes (hist_u16_kernel).
__global__ void __launch_bounds__(MT)
normal_metrics_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask, int HW,
                              long long n, int per, float* __restrict__ err, double* __restrict__ partial) {
    __shared__ double red[4];
    const float thr[5] = {5.0f, 7.5f, 11.25f, 22.5f, 30.0f};
    double s[MSTATS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long base = (long long)blockIdx.x * MT * per;
    for (int k = 0; k < per; ++k) {
        const long long i = base + (long long)k * MT + threadIdx.x;      // pixel index over B*HW
        if (i >= n) break;
        const long long b = i / HW, p = i - b * HW;
        float e = __uint_as_float(0xFFFFFFFFu);
        if (mask[i] > 0.f) {
            const float* pp = pred + b * 3 * HW + p;
            const float* gp = gt + b * 3 * HW + p;
            const float p0 = pp[0], p1 = pp[HW], p2 = pp[2 * (long long)HW], g0 = gp[0], g1 = gp[HW], g2 = gp[2 * (long long)HW];
            const float pn = fmaxf(sqrtf(p0 * p0 + p1 * p1 + p2 * p2), 1e-12f), gn = fmaxf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2), 1e-12f);   // F.normalize
            float d = (p0 / pn) * (g0 / gn) + (p1 / pn) * (g1 / gn) + (p2 / pn) * (g2 / gn);
            d = fminf(fmaxf(d, -1.0f), 1.0f);
            e = acosf(d) / 3.14159274f * 180.0f;
            s[0] += 1.0;
            s[1] += (double)e;
            s[2] += (double)e * (double)e;
#pragma unroll
            for (int t = 0; t < 5; ++t) s[3 + t] += (e < thr[t]) ? 1.0 : 0.0;
        }
        err[i] = e;
    }
#pragma unroll
    for (int q = 0; q < MSTATS; ++q) {
        const double v = wg_sum(s[q], red);
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * MSTATS + q] = v;
    }
}

// hist[65536] += counts of a 16-bit digit of the values' bit patterns: hi_filter < 0: the upper 16 bits; else the lower 16 bits of
// the values whose upper 16 bits equal hi_filter.  Non-negative floats order like their bit patterns, so two passes select any order
// statistic exactly; counts are integers, hence mergeable over batches and ranks in any order.  The all-ones sentinel is skipped.
__global__ void __launch_bounds__(MT) hist_u16_kernel(const float* __restrict__ vals, long long n, int hi_filter, unsigned* __restrict__ hist) {
    const long long i = (long long)blockIdx.x * MT + threadIdx.x;
    if (i >= n) return;
    const unsigned u = __float_as_uint(vals[i]);
    if (u == 0xFFFFFFFFu) return;
    if (hi_filter < 0) atomicAdd(&hist[u >> 16], 1u);
    else if ((int)(u >> 16) == hi_filter) atomicAdd(&hist[u & 0xFFFFu], 1u);
}

}  // namespace

extern "C" int vidc_normal_metrics(const float* pred, const float* gt, const float* mask, int B, int HW, float* err, double* stats,
                                   int accumulate, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(pred && gt && mask && err && stats && scratch, VIDC_ERR_NULL, "vidc_normal_metrics: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_normal_metrics: bad shape");
    const long long n = (long long)B * HW;
    const int per = 16;
    const int blocks = (int)((n + (long long)MT * per - 1) / ((long long)MT * per));
    hipStream_t st = vidc::as_stream(stream);
    hipLaunchKernelGGL(normal_metrics_partial_kernel, dim3(blocks), dim3(MT), 0, st, pred, gt, mask, HW, n, per, err, reinterpret_cast<double*>(scratch));
    VIDC_CHECK_LAUNCH("normal_metrics_partial_kernel");
    hipLaunchKernelGGL(depth_metrics_final_kernel, dim3(1), dim3(MT), 0, st, reinterpret_cast<const double*>(scratch), blocks, stats, accumulate);
    VIDC_CHECK_LAUNCH("depth_metrics_final_kernel");
    return VIDC_OK;
}

extern "C" int vidc_hist_u16(const float* vals, long long n, int hi_filter, uint32_t* hist, vidc_stream_t stream) {
    VIDC_REQUIRE(vals && hist, VIDC_ERR_NULL, "vidc_hist_u16: null pointer");
    VIDC_REQUIRE(n > 0 && hi_filter < 65536, VIDC_ERR_SHAPE, "vidc_hist_u16: bad arguments");
    hipLaunchKernelGGL(hist_u16_kernel, dim3((unsigned)((n + MT - 1) / MT)), dim3(MT), 0, vidc::as_stream(stream), vals, n, hi_filter, hist);
    VIDC_CHECK_LAUNCH("hist_u16_kernel");
    return VIDC_OK;
}

extern "C" size_t vidc_depth_metrics_scratch_bytes(long long n) {
    const long long blocks = (n + (long long)MT * 16 - 1) / ((long long)MT * 16);
    return (size_t)(blocks > 0 ? blocks : 1) * MSTATS * sizeof(double);
}

extern "C" int vidc_depth_metrics(const float* pred, const float* gt, long long n, double* stats, int accumulate, void* scratch,
                                  vidc_stream_t stream) {
    VIDC_REQUIRE(pred && gt && stats && scratch, VIDC_ERR_NULL, "vidc_depth_metrics: null pointer");
    VIDC_REQUIRE(n > 0, VIDC_ERR_SHAPE, "vidc_depth_metrics: n must be > 0");
    const int per = 16;
    const int blocks = (int)((n + (long long)MT * per - 1) / ((long long)MT * per));
    hipStream_t st = vidc::as_stream(stream);
    hipLaunchKernelGGL(depth_metrics_partial_kernel, dim3(blocks), dim3(MT), 0, st, pred, gt, n, per, reinterpret_cast<double*>(scratch));
    VIDC_CHECK_LAUNCH("depth_metrics_partial_kernel");
    hipLaunchKernelGGL(depth_metrics_final_kernel, dim3(1), dim3(MT), 0, st, reinterpret_cast<const double*>(scratch), blocks, stats, accumulate);
    VIDC_CHECK_LAUNCH("depth_metrics_final_kernel");
    return VIDC_OK;
}

extern "C" int vidc_depth_to_mm_u32(const float* depth, uint32_t* mm, long long n, vidc_stream_t stream) {
    VIDC_REQUIRE(depth && mm, VIDC_ERR_NULL, "vidc_depth_to_mm_u32: null pointer");
    VIDC_REQUIRE(n > 0, VIDC_ERR_SHAPE, "vidc_depth_to_mm_u32: n must be > 0");
    hipLaunchKernelGGL(depth_to_mm_kernel, dim3((unsigned)((n + MT - 1) / MT)), dim3(MT), 0, vidc::as_stream(stream), depth, mm, n);
    VIDC_CHECK_LAUNCH("depth_to_mm_kernel");
    return VIDC_OK;
}
