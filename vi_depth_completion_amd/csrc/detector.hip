// The two native kernels of the plane-mask head, re-designed for wave64:
//
//  * NMS (replaces csrc/cuda/nms.cu + the host loop of nms_cuda, and csrc/cpu/nms_cpu.cpp): the reference builds a 64x64-tiled IoU
//    bitmask on the GPU, copies ALL of it to the host (N * N/64 * 8 bytes) and resolves the greedy suppression there.  Here the
//    64-bit tile masks are one ballot-sized word per lane of a wave64, and the greedy resolution stays on the device: one wave walks
//    the score-ordered boxes 64 at a time -- the in-tile dependencies through lane broadcasts of the diagonal tile held in registers,
//    the cross-tile ones by OR-ing the kept rows into a per-lane `removed` word -- so only the kept indices ever leave the GPU.
//  * ROIAlign forward (replaces csrc/cuda/ROIAlign_cuda.cu:65-176 / csrc/cpu/ROIAlign_cpu.cpp): NHWC in (the conv engine's layout),
//    lanes run along channels, so the four bilinear taps of a sample are 256-byte coalesced reads; the sample geometry of a bin is
//    computed once per wave.
#include "common.h"
#include <cstdint>

namespace {

__device__ inline float box_iou(const float4 a, const float4 b) {            // nms.cu:13-21 (the "+1" convention)
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
    const float inter = width * height;
    const float sa = (a.z - a.x + 1.f) * (a.w - a.y + 1.f), sb = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
    return inter / (sa + sb - inter);
}

// mask[i][cb] bit j: sorted box i suppresses sorted box cb*64+j (only j > i within the diagonal tile; tiles left of it unused)
__global__ void __launch_bounds__(64)
nms_mask_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ order, int n, float thresh, int ge,
                unsigned long long* __restrict__ mask, int col_blocks) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    __shared__ float4 cbox[64];
    const int cj = cb * 64 + threadIdx.x;
    if (cj < n) cbox[threadIdx.x] = boxes[order[cj]];
    __syncthreads();
    const int ri = rb * 64 + threadIdx.x;
    if (ri >= n) return;
    const float4 me = boxes[order[ri]];
    const int cols = min(64, n - cb * 64);
    unsigned long long t = 0;
    for (int j = (rb == cb ? threadIdx.x + 1 : 0); j < cols; ++j) {
        const float v = box_iou(me, cbox[j]);
        if (ge ? (v >= thresh) : (v > thresh)) t |= 1ull << j;
    }
    mask[(size_t)ri * col_blocks + cb] = t;
}

// One wave.  Lane l owns the `removed` words of column tiles l, l+64, ... (RW of them).  keep_flag[original index] = 1 for survivors.
template <int RW>
__global__ void __launch_bounds__(64)
nms_reduce_kernel(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ order, int n, int col_blocks,
                  uint8_t* __restrict__ keep_flag) {
    const int lane = threadIdx.x;
    unsigned long long removed[RW];
#pragma unroll
    for (int w = 0; w < RW; ++w) removed[w] = 0;
    for (int blk = 0; blk < col_blocks; ++blk) {
        const int rows = min(64, n - blk * 64);
        // this tile's removed word so far (broadcast from its owner), and the diagonal tile: lane r holds row blk*64+r
        unsigned long long rem = 0;
#pragma unroll
        for (int w = 0; w < RW; ++w) {
            const unsigned long long v = __shfl(removed[w], blk & 63, 64);
            if ((blk >> 6) == w) rem = v;
        }
        const unsigned long long diag = lane < rows ? mask[(size_t)(blk * 64 + lane) * col_blocks + blk] : 0ull;
        unsigned long long kept = 0;
        for (int r = 0; r < rows; ++r) {                       // greedy, in score order; all lanes compute the same `rem`
            const unsigned long long row = __shfl(diag, r, 64);
            if (!((rem >> r) & 1ull)) { kept |= 1ull << r; rem |= row; }
        }
        if (lane < rows && ((kept >> lane) & 1ull)) keep_flag[order[blk * 64 + lane]] = 1;
        // kept rows suppress boxes of the later tiles: lane l ORs the words of its column tiles over ALL rows of this tile with
        // unconditional, independent loads (8 in flight per lane and word) and masks out the rows that were not kept
#pragma unroll
        for (int w = 0; w < RW; ++w) {
            const int cb = w * 64 + lane;
            if (cb <= blk || cb >= col_blocks) continue;
            const unsigned long long* mcol = mask + (size_t)(blk * 64) * col_blocks + cb;
            unsigned long long acc = 0;
            int r = 0;
            for (; r + 8 <= rows; r += 8) {
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = mcol[(size_t)(r + u) * col_blocks];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc |= ((kept >> (r + u)) & 1ull) ? v[u] : 0ull;
            }
            for (; r < rows; ++r) acc |= ((kept >> r) & 1ull) ? mcol[(size_t)r * col_blocks] : 0ull;
            removed[w] |= acc;
        }
    }
}

// keep[0..count) = the original indices with keep_flag set, ascending (the reference returns them sorted); one workgroup
__global__ void __launch_bounds__(256)
nms_compact_kernel(const uint8_t* __restrict__ keep_flag, int n, int32_t* __restrict__ keep, int32_t* __restrict__ n_keep) {
    __shared__ int wsum[4];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool f = i < n && keep_flag[i];
        const unsigned long long b = __ballot(f);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (f) keep[off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_keep = base;
}

// ---- segmented NMS: S independent box lists (one per (image, pyramid level), say), each already in descending score order, in one
//      launch set.  Segment s = boxes[off[s] .. off[s] + n[s]); keep[off[s] + i] = i-th survivor (index inside the segment, ascending =
//      score order), n_keep[s] their count.  mask scratch: [S][max_n][cb_max] words. -----------------------------------------------------
__global__ void __launch_bounds__(64)
nms_seg_mask_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_n, float thresh, int ge,
                    unsigned long long* __restrict__ mask, int max_n, int cb_max) {
    const int s = blockIdx.z, n = seg_n[s], rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
    const float4* bx = boxes + seg_off[s];
    __shared__ float4 cbox[64];
    const int cj = cb * 64 + threadIdx.x;
    if (cj < n) cbox[threadIdx.x] = bx[cj];
    __syncthreads();
    const int ri = rb * 64 + threadIdx.x;
    if (ri >= n) return;
    const float4 me = bx[ri];
    const int cols = min(64, n - cb * 64);
    unsigned long long t = 0;
    for (int j = (rb == cb ? threadIdx.x + 1 : 0); j < cols; ++j) {
        const float v = box_iou(me, cbox[j]);
        if (ge ? (v >= thresh) : (v > thresh)) t |= 1ull << j;
    }
    mask[((size_t)s * max_n + ri) * cb_max + cb] = t;
}

// one wave per segment (n <= 4096): the greedy walk of nms_reduce_kernel<1>, survivors written straight to keep[] in order
__global__ void __launch_bounds__(64)
nms_seg_reduce_kernel(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_n,
                      int max_n, int cb_max, int max_keep, int32_t* __restrict__ keep, int32_t* __restrict__ n_keep) {
    const int s = blockIdx.x, n = seg_n[s], lane = threadIdx.x;
    const int col_blocks = (n + 63) / 64;
    const unsigned long long* m = mask + (size_t)s * max_n * cb_max;
    int32_t* kp = keep + seg_off[s];
    unsigned long long removed = 0;                       // lane l: the removed word of column tile l
    int count = 0;
    for (int blk = 0; blk < col_blocks && count < max_keep; ++blk) {     // later boxes never change earlier decisions: stop when full
        const int rows = min(64, n - blk * 64);
        unsigned long long rem = __shfl(removed, blk, 64);
        const unsigned long long diag = lane < rows ? m[(size_t)(blk * 64 + lane) * cb_max + blk] : 0ull;
        unsigned long long kept = 0;
        for (int r = 0; r < rows; ++r) {
            const unsigned long long row = __shfl(diag, r, 64);
            if (!((rem >> r) & 1ull)) { kept |= 1ull << r; rem |= row; }
        }
        if (count + __popcll(kept) > max_keep) {          // keep only the first max_keep - count survivors of this tile
            int room = max_keep - count;
            unsigned long long t = kept, first = 0;
            while (room-- > 0) { const unsigned long long low = t & (~t + 1ull); first |= low; t ^= low; }
            kept = first;
        }
        if (lane < rows && ((kept >> lane) & 1ull)) kp[count + __popcll(kept & ((1ull << lane) - 1ull))] = blk * 64 + lane;
        count += __popcll(kept);
        if (lane > blk && lane < col_blocks) {
            const unsigned long long* mcol = m + (size_t)(blk * 64) * cb_max + lane;
            unsigned long long acc = 0;
            for (int r = 0; r < rows; ++r) acc |= ((kept >> r) & 1ull) ? mcol[(size_t)r * cb_max] : 0ull;
            removed |= acc;
        }
    }
    if (lane == 0) n_keep[s] = count;
}

// ---- ROIAlign forward, NHWC --------------------------------------------------------------------------------------------
// grid (pooled_w * pooled_h, K); block = 64 * CW threads: lanes along channels
__global__ void __launch_bounds__(256)
roi_align_fwd_kernel(const float* __restrict__ x, const float* __restrict__ rois, float* __restrict__ y, int C, int H, int W, int ldx,
                     int PH, int PW, float scale, int sampling_ratio) {
    const int k = blockIdx.y, ph = blockIdx.x / PW, pw = blockIdx.x - ph * PW;
    const float* r = rois + (size_t)k * 5;
    const int b = (int)r[0];
    const float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;      // "Do not use rounding" (ROIAlign_cpu.cpp:137-141)
    const float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);                               // malformed ROIs forced to 1x1
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    const float count = (float)(gh * gw);
    const float* xb = x + (size_t)b * H * W * ldx;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            float yy = sh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                float xx = sw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
                float yv = yy;
                if (yv < -1.0f || yv > (float)H || xx < -1.0f || xx > (float)W) continue;         // sample outside: contributes 0
                if (yv <= 0.f) yv = 0.f;
                if (xx <= 0.f) xx = 0.f;
                int yl = (int)yv, xl = (int)xx, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; yv = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
                const float ly = yv - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                // the reference's expression order: w1*p1 + w2*p2 + w3*p3 + w4*p4, added to the running sum
                const float v = w1 * xb[((size_t)yl * W + xl) * ldx + c] + w2 * xb[((size_t)yl * W + xh) * ldx + c] +
                                w3 * xb[((size_t)yh * W + xl) * ldx + c] + w4 * xb[((size_t)yh * W + xh) * ldx + c];
                acc += v;
            }
        }
        y[(((size_t)k * PH + ph) * PW + pw) * C + c] = acc / count;
    }
}

}  // namespace

extern "C" size_t vidc_nms_scratch_bytes(int n) {
    const size_t cb = (size_t)(n + 63) / 64;
    return (size_t)n * cb * sizeof(unsigned long long) + (size_t)n + 64;
}

extern "C" int vidc_nms(const float* boxes_xyxy, const int32_t* order, int n, float threshold, int inclusive, int32_t* keep,
                        int32_t* n_keep, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(keep && n_keep, VIDC_ERR_NULL, "vidc_nms: null output pointer");
    hipStream_t st = vidc::as_stream(stream);
    if (n == 0) {
        VIDC_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), st));
        return VIDC_OK;
    }
    VIDC_REQUIRE(boxes_xyxy && order && scratch, VIDC_ERR_NULL, "vidc_nms: null pointer");
    VIDC_REQUIRE(n > 0 && n <= 64 * 64 * 4, VIDC_ERR_SHAPE, "vidc_nms: n = %d out of range (1..16384)", n);
    VIDC_REQUIRE(((size_t)boxes_xyxy & 15) == 0, VIDC_ERR_SHAPE, "vidc_nms: boxes must be 16-byte aligned");
    const int cb = (n + 63) / 64;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(scratch);
    uint8_t* flag = reinterpret_cast<uint8_t*>(mask + (size_t)n * cb);
    VIDC_HIP(hipMemsetAsync(flag, 0, (size_t)n, st));
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, st, reinterpret_cast<const float4*>(boxes_xyxy), order, n, threshold, inclusive,
                       mask, cb);
    VIDC_CHECK_LAUNCH("nms_mask_kernel");
    if (cb <= 64)       hipLaunchKernelGGL(nms_reduce_kernel<1>, dim3(1), dim3(64), 0, st, mask, order, n, cb, flag);
    else if (cb <= 128) hipLaunchKernelGGL(nms_reduce_kernel<2>, dim3(1), dim3(64), 0, st, mask, order, n, cb, flag);
    else                hipLaunchKernelGGL(nms_reduce_kernel<4>, dim3(1), dim3(64), 0, st, mask, order, n, cb, flag);
    VIDC_CHECK_LAUNCH("nms_reduce_kernel");
    hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(256), 0, st, flag, n, keep, n_keep);
    VIDC_CHECK_LAUNCH("nms_compact_kernel");
    return VIDC_OK;
}

extern "C" size_t vidc_nms_segmented_scratch_bytes(int n_segments, int max_n) {
    return (size_t)n_segments * max_n * ((max_n + 63) / 64) * sizeof(unsigned long long) + 64;
}

extern "C" int vidc_nms_segmented(const float* boxes_xyxy, const int32_t* seg_offsets, const int32_t* seg_counts, int n_segments, int max_n,
                                  float threshold, int inclusive, int max_keep, int32_t* keep, int32_t* n_keep, void* scratch,
                                  vidc_stream_t stream) {
    VIDC_REQUIRE(boxes_xyxy && seg_offsets && seg_counts && keep && n_keep && scratch, VIDC_ERR_NULL, "vidc_nms_segmented: null pointer");
    VIDC_REQUIRE(n_segments > 0 && max_n > 0 && max_n <= 4096, VIDC_ERR_SHAPE, "vidc_nms_segmented: max_n = %d out of range (1..4096)", max_n);
    VIDC_REQUIRE(((size_t)boxes_xyxy & 15) == 0, VIDC_ERR_SHAPE, "vidc_nms_segmented: boxes must be 16-byte aligned");
    hipStream_t st = vidc::as_stream(stream);
    const int cb = (max_n + 63) / 64;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(scratch);
    hipLaunchKernelGGL(nms_seg_mask_kernel, dim3(cb, cb, n_segments), dim3(64), 0, st, reinterpret_cast<const float4*>(boxes_xyxy), seg_offsets,
                       seg_counts, threshold, inclusive, mask, max_n, cb);
    VIDC_CHECK_LAUNCH("nms_seg_mask_kernel");
    hipLaunchKernelGGL(nms_seg_reduce_kernel, dim3(n_segments), dim3(64), 0, st, mask, seg_offsets, seg_counts, max_n, cb, max_keep > 0 ? max_keep : max_n, keep, n_keep);
    VIDC_CHECK_LAUNCH("nms_seg_reduce_kernel");
    return VIDC_OK;
}

extern "C" int vidc_roi_align_forward(const float* x_nhwc, const float* rois, float* y, int K, int C, int H, int W, int ldx, int pooled_h,
                                      int pooled_w, float spatial_scale, int sampling_ratio, vidc_stream_t stream) {
    if (K == 0) return VIDC_OK;
    VIDC_REQUIRE(x_nhwc && rois && y, VIDC_ERR_NULL, "vidc_roi_align_forward: null pointer");
    VIDC_REQUIRE(K > 0 && C > 0 && H > 0 && W > 0 && ldx >= C && pooled_h > 0 && pooled_w > 0, VIDC_ERR_SHAPE, "vidc_roi_align_forward: bad shape");
    const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(pooled_h * pooled_w, K), dim3(threads), 0, vidc::as_stream(stream), x_nhwc, rois, y, C, H, W, ldx,
                       pooled_h, pooled_w, spatial_scale, sampling_ratio);
    VIDC_CHECK_LAUNCH("roi_align_fwd_kernel");
    return VIDC_OK;
}
