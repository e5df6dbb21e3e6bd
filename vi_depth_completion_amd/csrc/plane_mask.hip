// Plane-mask detector: everything of the reference's R-101-FPN Mask R-CNN inference path that is not a convolution
// (those run on the conv engine, networks/plane_mask_rcnn.py).  All shapes are static (R proposal / detection slots per image, counts
// kept in device memory), nothing synchronises with the host, and every kernel processes the whole batch.
//
//   det_stem_im2col      demo/predictor.py:101-118,143-144 (uint8 cast, BGR, x255, -mean) + zero pad to /32 + the 7x7/s2 stem's im2col
//   upsample_nearest2x   modeling/backbone/fpn.py:66-72 (top-down path)
//   rpn_topk_decode      modeling/rpn/inference.py:74-110 (sigmoid, top-k sorted, anchors, BoxCoder.decode, clip)
//   rpn_select           modeling/rpn/inference.py:103-108 (first post_nms_top_n survivors per level) + :148-190 (top-k over levels)
//   roi_align_fpn        modeling/poolers.py:11-122 (LevelMapper + per-level ROIAlign), csrc/cpu/ROIAlign_cpu.cpp
//   det_candidates/_select   modeling/roi_heads/box_head/inference.py:47-146
//   mask_paste           modeling/roi_heads/mask_head/inference.py:27-49 (sigmoid, own class) + :86-150 (Masker(0.5, padding 1))
//   instance_map         demo/predictor.py:201-220, 253-323 (confidence filter, biggest 4-connected component, >= 5 % area, ids by size)
#include "common.h"
#include <cstdint>

namespace {

constexpr float BBOX_XFORM_CLIP = 4.135166556742356f;      // log(1000 / 16), box_coder.py:19

// ---- dense helpers ------------------------------------------------------------------------------------------------------------
// cols[b][oy][ox][k], k = (kh*7 + kw)*3 + c (c = B,G,R), 147 real + 13 zero columns = 160 = 5 K units of the conv kernel.
__global__ void __launch_bounds__(256)
det_stem_im2col_kernel(const float* __restrict__ img, float* __restrict__ cols, int B, int H, int W, int Hp, int Wp, float mb, float mg, float mr) {
    const int Ho = Hp / 2, Wo = Wp / 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (pixel, tap)
    if (idx >= (long long)B * Ho * Wo * 54) return;
    const int tap = (int)(idx % 54);
    long long t = idx / 54;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float* dst = cols + ((size_t)(b * Ho + oy) * Wo + ox) * 160;
    if (tap >= 49) {                                         // taps 49..53 write the 13 zero columns (3 + 3 + 3 + 3 + 1)
        const int k0 = 147 + (tap - 49) * 3;
        for (int k = k0; k < min(k0 + 3, 160); ++k) dst[k] = 0.f;
        return;
    }
    const int kh = tap / 7, kw = tap - kh * 7;
    const int iy = oy * 2 - 3 + kh, ix = ox * 2 - 3 + kw;
    float v[3] = {0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {      // inside the real image; the pad rows/cols of the /32 padding are zeros
        const float mean[3] = {mb, mg, mr};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = img[((size_t)(b * 3 + (2 - c)) * H + iy) * W + ix];          // BGR <- RGB planes
            const float u8 = floorf(fminf(fmaxf(255.0f * x, 0.f), 255.f));                // np.asarray(255. * x, dtype=np.uint8)
            v[c] = (u8 / 255.0f) * 255.0f - mean[c];                                     // ToTensor, x255, Normalize(mean, 1)
        }
    }
    dst[tap * 3 + 0] = v[0]; dst[tap * 3 + 1] = v[1]; dst[tap * 3 + 2] = v[2];
}

__global__ void __launch_bounds__(256)
upsample_nearest2x_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int h, int w, int C, int ldx, int ldy) {
    const int q = C / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * 2 * h * 2 * w * q) return;
    const int c = (int)(idx % q) * 4;
    long long t = idx / q;
    const int ox = (int)(t % (2 * w)); t /= 2 * w;
    const int oy = (int)(t % (2 * h));
    const int b = (int)(t / (2 * h));
    *reinterpret_cast<float4*>(&y[((size_t)(b * 2 * h + oy) * 2 * w + ox) * ldy + c]) =
        *reinterpret_cast<const float4*>(&x[((size_t)(b * h + oy / 2) * w + ox / 2) * ldx + c]);
}

// ---- sorting: (key descending, index ascending) bitonic network over n = power of two items in LDS --------------------------------
__device__ inline bool before(float ka, int ia, float kb, int ib) { return ka > kb || (ka == kb && ia < ib); }

__device__ inline void bitonic_sort_desc(float* key, int* idx, int n) {
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;                     // this pair sorts "first element before second"
                    const float ka = key[i], kb = key[p];
                    const int ia = idx[i], ib = idx[p];
                    if (before(kb, ib, ka, ia) == up) { key[i] = kb; key[p] = ka; idx[i] = ib; idx[p] = ia; }
                }
            }
        }
    __syncthreads();
}

__device__ inline void decode_clip(const float d[4], const float a[4], float wx, float wy, float ww, float wh, float img_w, float img_h,
                                   float out[4]) {
    const float widths = a[2] - a[0] + 1.f, heights = a[3] - a[1] + 1.f;
    const float cx = a[0] + 0.5f * widths, cy = a[1] + 0.5f * heights;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw = fminf(d[2] / ww, BBOX_XFORM_CLIP), dh = fminf(d[3] / wh, BBOX_XFORM_CLIP);
    const float pcx = dx * widths + cx, pcy = dy * heights + cy;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    out[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.f), img_w - 1.f);
    out[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.f), img_h - 1.f);
    out[2] = fminf(fmaxf(pcx + 0.5f * pw - 1.f, 0.f), img_w - 1.f);
    out[3] = fminf(fmaxf(pcy + 0.5f * ph - 1.f, 0.f), img_h - 1.f);
}

struct Anchors { float a[3][4]; };

// One workgroup per image: sorts the A*h*w objectness values of one level (LDS, n2 = next power of two), decodes the top k.
// map: [B][h][w][ld], channels 0..A-1 = objectness logits, A..5A-1 = box deltas in (anchor, coordinate) order.
__device__ __forceinline__ void
rpn_topk_decode_body(const float* __restrict__ map, int h, int w, int ld, int A, int stride, const Anchors& cell, int n2, int k, float img_w,
                     float img_h, float* __restrict__ boxes, float* __restrict__ scores, long long out_stride) {
    extern __shared__ unsigned char lds[];
    float* key = reinterpret_cast<float*>(lds);
    int* idx = reinterpret_cast<int*>(lds) + n2;
    const int b = blockIdx.x, n = A * h * w;
    const float* mb = map + (size_t)b * h * w * ld;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        float v = -1.f;                                              // padding sorts last (sigmoid > 0)
        if (i < n) { const int pos = i / A, a = i - pos * A; v = 1.f / (1.f + expf(-mb[(size_t)pos * ld + a])); }
        key[i] = v; idx[i] = i;
    }
    bitonic_sort_desc(key, idx, n2);
    for (int r = threadIdx.x; r < k; r += blockDim.x) {
        const int i = idx[r], pos = i / A, a = i - pos * A, y = pos / w, x = pos - y * w;
        const float* dp = mb + (size_t)pos * ld + A + a * 4;
        const float d[4] = {dp[0], dp[1], dp[2], dp[3]};
        const float sx = (float)(x * stride), sy = (float)(y * stride);
        const float an[4] = {cell.a[a][0] + sx, cell.a[a][1] + sy, cell.a[a][2] + sx, cell.a[a][3] + sy};
        float o[4];
        decode_clip(d, an, 1.f, 1.f, 1.f, 1.f, img_w, img_h, o);
        float* ob = boxes + (size_t)b * out_stride * 4 + (size_t)r * 4;
        ob[0] = o[0]; ob[1] = o[1]; ob[2] = o[2]; ob[3] = o[3];
        scores[(size_t)b * out_stride + r] = key[r];
    }
}

// The same result for n > 1024 >= k without sorting everything: a 3-level radix select (11 + 11 + 10 bits of the float pattern; sigmoid
// outputs are positive, so the bit patterns order like the values) finds the k-th largest value T exactly, everything above T plus the
// lowest-index `need` elements equal to T are compacted (<= 1024 items), and only those are sorted.  16 K keys: ~50 us instead of ~300.
__device__ __forceinline__ void
rpn_topk_select_decode_body(const float* __restrict__ map, int h, int w, int ld, int A, int stride, const Anchors& cell, int k, float img_w,
                            float img_h, float* __restrict__ boxes, float* __restrict__ scores, long long out_stride) {
    extern __shared__ unsigned char lds[];
    const int b = blockIdx.x, n = A * h * w;
    unsigned* key = reinterpret_cast<unsigned*>(lds);                  // [n]
    int* hist = reinterpret_cast<int*>(lds) + n;                       // [2048]
    float* skey = reinterpret_cast<float*>(hist + 2048);               // [1024]
    int* sidx = reinterpret_cast<int*>(skey + 1024);                   // [1024]
    __shared__ int s_bin, s_need, s_gt, wcnt[16], s_base;
    const float* mb = map + (size_t)b * h * w * ld;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int pos = i / A, a = i - pos * A;
        key[i] = __float_as_uint(1.f / (1.f + expf(-mb[(size_t)pos * ld + a])));
    }
    if (threadIdx.x == 0) { s_need = k; s_gt = 0; s_base = 0; }
    unsigned prefix = 0, mask = 0;
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int shift = shifts[lvl], bins = 1 << nbits[lvl];
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if ((key[i] & mask) == prefix) atomicAdd(&hist[(key[i] >> shift) & (bins - 1)], 1);
        __syncthreads();
        if (wv == 0) {                                               // one wave: lane l owns bins [top - 32 l - 31, top - 32 l] from the top
            const int per = bins / 64;
            int sum = 0;
            for (int j = 0; j < per; ++j) sum += hist[bins - 1 - (lane * per + j)];
            int incl = sum;                                          // inclusive scan over the lanes (from the top bin down)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
            const int need = s_need;
            const unsigned long long hit = __ballot(incl >= need);
            const int owner = __ffsll((long long)hit) - 1;           // first lane whose cumulative count reaches `need`
            if (lane == owner) {
                int above = incl - sum, bin = 0;
                for (int j = 0; j < per; ++j) {
                    bin = bins - 1 - (lane * per + j);
                    if (above + hist[bin] >= need) break;
                    above += hist[bin];
                }
                s_bin = bin;
                s_need = need - above;                               // how many elements of this bin (and, at the end, equal to T) are wanted
            }
        }
        __syncthreads();
        prefix |= (unsigned)s_bin << shift;
        mask |= (unsigned)(bins - 1) << shift;
    }
    const unsigned T = prefix;
    const int need_eq = s_need, n_gt = k - need_eq;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) { skey[i] = -1.f; sidx[i] = 0x7fffffff; }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {                           // in index order: ties at T keep the lowest indices
        const int i = i0 + threadIdx.x;
        const unsigned v = i < n ? key[i] : 0u;
        if (i < n && v > T) { const int p = atomicAdd(&s_gt, 1); skey[p] = __uint_as_float(v); sidx[p] = i; }
        const bool eq = i < n && v == T;
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int rank = s_base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int q = 0; q < wv; ++q) rank += wcnt[q];
        if (eq && rank < need_eq) { skey[n_gt + rank] = __uint_as_float(v); sidx[n_gt + rank] = i; }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += wcnt[q]; s_base += t; }
        __syncthreads();
    }
    bitonic_sort_desc(skey, sidx, 1024);
    for (int r = threadIdx.x; r < k; r += blockDim.x) {
        const int i = sidx[r], pos = i / A, a = i - pos * A, y = pos / w, x = pos - y * w;
        const float* dp = mb + (size_t)pos * ld + A + a * 4;
        const float d[4] = {dp[0], dp[1], dp[2], dp[3]};
        const float sx = (float)(x * stride), sy = (float)(y * stride);
        const float an[4] = {cell.a[a][0] + sx, cell.a[a][1] + sy, cell.a[a][2] + sx, cell.a[a][3] + sy};
        float o[4];
        decode_clip(d, an, 1.f, 1.f, 1.f, 1.f, img_w, img_h, o);
        float* ob = boxes + (size_t)b * out_stride * 4 + (size_t)r * 4;
        ob[0] = o[0]; ob[1] = o[1]; ob[2] = o[2]; ob[3] = o[3];
        scores[(size_t)b * out_stride + r] = skey[r];
    }
}

__global__ void __launch_bounds__(1024)
rpn_topk_decode_kernel(const float* __restrict__ map, int h, int w, int ld, int A, int stride, Anchors cell, int n2, int k, float img_w,
                       float img_h, float* __restrict__ boxes, float* __restrict__ scores, long long out_stride) {
    rpn_topk_decode_body(map, h, w, ld, A, stride, cell, n2, k, img_w, img_h, boxes, scores, out_stride);
}
__global__ void __launch_bounds__(1024)
rpn_topk_select_decode_kernel(const float* __restrict__ map, int h, int w, int ld, int A, int stride, Anchors cell, int k, float img_w,
                              float img_h, float* __restrict__ boxes, float* __restrict__ scores, long long out_stride) {
    rpn_topk_select_decode_body(map, h, w, ld, A, stride, cell, k, img_w, img_h, boxes, scores, out_stride);
}
// All pyramid levels in one launch: grid (B, n_levels); every workgroup runs the path its level's size asks for.
struct LevelDesc { const float* map; int h, w, stride, k, n2, off; Anchors cell; };
struct LevelSet { LevelDesc lv[5]; };
__global__ void __launch_bounds__(1024)
rpn_topk_levels_kernel(LevelSet ls, int ld, int A, float img_w, float img_h, float* __restrict__ boxes, float* __restrict__ scores,
                       long long out_stride) {
    const LevelDesc& d = ls.lv[blockIdx.y];
    float* bo = boxes + (size_t)d.off * 4;
    float* so = scores + d.off;
    if (A * d.h * d.w > 1024 && d.k <= 1024) rpn_topk_select_decode_body(d.map, d.h, d.w, ld, A, d.stride, d.cell, d.k, img_w, img_h, bo, so, out_stride);
    else                                     rpn_topk_decode_body(d.map, d.h, d.w, ld, A, d.stride, d.cell, d.n2, d.k, img_w, img_h, bo, so, out_stride);
}

struct Levels { int n_levels; int off[8]; };      // off[l]..off[l+1]: slots of level l inside the per-image candidate arrays

// One workgroup (256 threads) per image: the first `per_level` NMS survivors of every level (they are in score order), then the best
// `total` of them over all levels (stable: ties keep level-major order).
__global__ void __launch_bounds__(256)
rpn_select_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ keep,
                  const int32_t* __restrict__ n_keep, Levels lv, int slots, int per_level, int total, float* __restrict__ props,
                  float* __restrict__ prop_scores, int32_t* __restrict__ n_props) {
    __shared__ float key[512];
    __shared__ int idx[512];
    __shared__ int src[512];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) { key[i] = -1.f; idx[i] = i; src[i] = -1; }
    __syncthreads();
    int base = 0;
    for (int l = 0; l < lv.n_levels; ++l) {
        const int nk = min(n_keep[b * lv.n_levels + l], per_level);
        for (int i = threadIdx.x; i < nk; i += blockDim.x) {
            const int s = lv.off[l] + keep[(size_t)b * slots + lv.off[l] + i];
            key[base + i] = scores[(size_t)b * slots + s];
            src[base + i] = s;
        }
        base += nk;                                               // uniform: n_keep is read by every thread
    }
    bitonic_sort_desc(key, idx, 512);
    const int n = min(base, total);
    for (int r = threadIdx.x; r < total; r += blockDim.x) {
        float o[4] = {0.f, 0.f, 0.f, 0.f}, sc = 0.f;
        if (r < n) {
            const int s = src[idx[r]];
            const float* bp = boxes + ((size_t)b * slots + s) * 4;
            o[0] = bp[0]; o[1] = bp[1]; o[2] = bp[2]; o[3] = bp[3];
            sc = key[r];
        }
        float* pp = props + ((size_t)b * total + r) * 4;
        pp[0] = o[0]; pp[1] = o[1]; pp[2] = o[2]; pp[3] = o[3];
        prop_scores[(size_t)b * total + r] = sc;
    }
    if (threadIdx.x == 0) n_props[b] = n;
}

// ---- ROIAlign over the pyramid ---------------------------------------------------------------------------------------------------
struct Pyramid { const float* x[4]; int h[4], w[4]; float scale[4]; };

__global__ void __launch_bounds__(256)
roi_align_fpn_kernel(Pyramid py, int C, const float* __restrict__ boxes, int R, int P, int sampling_ratio, float* __restrict__ y) {
    const int k = blockIdx.y, ph = blockIdx.x / P, pw = blockIdx.x - ph * P;
    const int b = k / R;
    const float* r = boxes + (size_t)k * 4;
    // LevelMapper (poolers.py:31-40): floor(4 + log2(sqrt(area) / 224 + 1e-6)) clamped to [2, 5]
    const float area = (r[2] - r[0] + 1.f) * (r[3] - r[1] + 1.f);
    const float lf = floorf(4.f + log2f(sqrtf(area) / 224.f + 1e-6f));
    const int lvl = (int)fminf(fmaxf(lf, 2.f), 5.f) - 2;
    const int H = py.h[lvl], W = py.w[lvl];
    const float scale = py.scale[lvl];
    const float sw = r[0] * scale, sh = r[1] * scale, ew = r[2] * scale, eh = r[3] * scale;
    const float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = (float)(gh * gw);
    const float* xb = py.x[lvl] + (size_t)b * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float yy = sh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                float xx = sw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
                float yv = yy;
                if (yv < -1.0f || yv > (float)H || xx < -1.0f || xx > (float)W) continue;
                if (yv <= 0.f) yv = 0.f;
                if (xx <= 0.f) xx = 0.f;
                int yl = (int)yv, xl = (int)xx, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; yv = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
                const float ly = yv - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                acc += w1 * xb[((size_t)yl * W + xl) * C + c] + w2 * xb[((size_t)yl * W + xh) * C + c] +
                       w3 * xb[((size_t)yh * W + xl) * C + c] + w4 * xb[((size_t)yh * W + xh) * C + c];
            }
        }
        y[(((size_t)k * P + ph) * P + pw) * C + c] = acc / count;
    }
}

// ---- box head post-processing ---------------------------------------------------------------------------------------------------
// One workgroup (64 threads) per image.  head: [B*R][ld] = 2 class logits, then 8 box deltas (class-major).  Candidates = proposals
// whose plane-class probability exceeds score_thresh, sorted by score (the order NMS walks them in); the rest of the R slots get
// far-away unit boxes that overlap nothing.
__global__ void __launch_bounds__(64)
det_candidates_kernel(const float* __restrict__ head, int ld, const float* __restrict__ props, const int32_t* __restrict__ n_props, int R,
                      float img_w, float img_h, float score_thresh, float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
                      int32_t* __restrict__ cand_src, int32_t* __restrict__ n_cand) {
    __shared__ float key[64];
    __shared__ int idx[64];
    __shared__ float bx[64][4];
    const int b = blockIdx.x, t = threadIdx.x;
    float sc = -1.f;
    if (t < R && t < n_props[b]) {
        const float* hp = head + ((size_t)b * R + t) * ld;
        const float m = fmaxf(hp[0], hp[1]);
        const float e0 = expf(hp[0] - m), e1 = expf(hp[1] - m);
        const float p1 = e1 / (e0 + e1);                              // F.softmax(class_logits, -1)[:, 1]
        const float* pp = props + ((size_t)b * R + t) * 4;
        const float d[4] = {hp[2 + 4], hp[2 + 5], hp[2 + 6], hp[2 + 7]}, an[4] = {pp[0], pp[1], pp[2], pp[3]};
        float o[4];
        decode_clip(d, an, 10.f, 10.f, 5.f, 5.f, img_w, img_h, o);
        bx[t][0] = o[0]; bx[t][1] = o[1]; bx[t][2] = o[2]; bx[t][3] = o[3];
        if (p1 > score_thresh) sc = p1;
    }
    key[t] = sc; idx[t] = t;
    bitonic_sort_desc(key, idx, 64);
    const unsigned long long valid = __ballot(key[t] > 0.f);
    if (t < R) {
        const bool ok = key[t] > 0.f;
        const int s = idx[t];
        float* cb = cand_boxes + ((size_t)b * R + t) * 4;
        if (ok) { cb[0] = bx[s][0]; cb[1] = bx[s][1]; cb[2] = bx[s][2]; cb[3] = bx[s][3]; }
        else    { cb[0] = -1.0e6f - 10.f * t; cb[1] = -1.0e6f; cb[2] = cb[0]; cb[3] = cb[1]; }
        cand_scores[(size_t)b * R + t] = ok ? key[t] : 0.f;
        cand_src[(size_t)b * R + t] = ok ? s : -1;
    }
    if (t == 0) n_cand[b] = __popcll(valid);
}

// One workgroup (64 threads) per image: NMS survivors among the real candidates, in ascending proposal order (the reference's CPU
// NMS returns ascending indices and filter_results keeps that order).
__global__ void __launch_bounds__(64)
det_select_kernel(const float* __restrict__ cand_boxes, const float* __restrict__ cand_scores, const int32_t* __restrict__ cand_src,
                  const int32_t* __restrict__ n_cand, const int32_t* __restrict__ keep, const int32_t* __restrict__ n_keep, int R,
                  float* __restrict__ det_boxes, float* __restrict__ det_scores, int32_t* __restrict__ n_det) {
    __shared__ float key[64];
    __shared__ int idx[64];
    const int b = blockIdx.x, t = threadIdx.x;
    int c = -1;
    if (t < n_keep[b]) { const int kk = keep[(size_t)b * R + t]; if (kk < n_cand[b]) c = kk; }
    key[t] = c >= 0 ? -(float)cand_src[(size_t)b * R + c] : -1.0e9f;          // descending -src = ascending proposal index
    idx[t] = c;
    bitonic_sort_desc(key, idx, 64);
    const unsigned long long valid = __ballot(idx[t] >= 0);
    if (t < R) {
        const int cc = idx[t];
        float* db = det_boxes + ((size_t)b * R + t) * 4;
        if (cc >= 0) {
            const float* cb = cand_boxes + ((size_t)b * R + cc) * 4;
            db[0] = cb[0]; db[1] = cb[1]; db[2] = cb[2]; db[3] = cb[3];
            det_scores[(size_t)b * R + t] = cand_scores[(size_t)b * R + cc];
        } else {
            db[0] = db[1] = db[2] = db[3] = 0.f;
            det_scores[(size_t)b * R + t] = 0.f;
        }
    }
    if (t == 0) n_det[b] = __popcll(valid);
}

// ---- masks ------------------------------------------------------------------------------------------------------------------------
// logits: [B*R][M/2][M/2 * 4][ld]: the 2x2 transposed conv ran as a 1x1 conv to 4*256 channels, so the 28x28 mask pixel (py, px) sits at
// [py/2][(px/2)*4 + (py&1)*2 + (px&1)]; channel `cls` is the detection's own class.  grid = (ceil(W*H/256), R, B).
__global__ void __launch_bounds__(256)
mask_paste_kernel(const float* __restrict__ logits, int ld, int cls, int M, const float* __restrict__ dets, const int32_t* __restrict__ n_det,
                  int R, int H, int W, float thresh, uint8_t* __restrict__ pasted) {
    const int b = blockIdx.z, k = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    uint8_t* out = pasted + ((size_t)(b * R + k) * H) * W;
    if (k >= n_det[b]) { out[p] = 0; return; }
    const float* d = dets + ((size_t)b * R + k) * 4;
    const int MP = M + 2;                                              // padding = 1 on each side
    const float scale = (float)MP / (float)M;
    const float w_half = (d[2] - d[0]) * .5f * scale, h_half = (d[3] - d[1]) * .5f * scale;
    const float xc = (d[2] + d[0]) * .5f, yc = (d[3] + d[1]) * .5f;
    const int bx0 = (int)(xc - w_half), by0 = (int)(yc - h_half), bx1 = (int)(xc + w_half), by1 = (int)(yc + h_half);   // .to(torch.int32)
    const int bw = max(bx1 - bx0 + 1, 1), bh = max(by1 - by0 + 1, 1);
    const int y = p / W, x = p - y * W;
    uint8_t v = 0;
    if (x >= max(bx0, 0) && x < min(bx1 + 1, W) && y >= max(by0, 0) && y < min(by1 + 1, H)) {
        // F.interpolate(padded, size=(bh, bw), mode='bilinear', align_corners=False) at (y - by0, x - bx0)
        const float sy = fmaxf(((float)MP / (float)bh) * ((float)(y - by0) + 0.5f) - 0.5f, 0.f);
        const float sx = fmaxf(((float)MP / (float)bw) * ((float)(x - bx0) + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < MP - 1 ? 1 : 0), x1 = x0 + (x0 < MP - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* lg = logits + (size_t)(b * R + k) * (M / 2) * (M / 2) * 4 * ld + cls;
        auto prob = [&](int py, int px) -> float {                    // padded coordinates; the border is zero
            if (py < 1 || py > M || px < 1 || px > M) return 0.f;
            py -= 1; px -= 1;
            const float l = lg[((size_t)(py >> 1) * (M / 2) * 4 + (size_t)(px >> 1) * 4 + (py & 1) * 2 + (px & 1)) * ld];
            return 1.f / (1.f + expf(-l));
        };
        const float val = hy * (hx * prob(y0, x0) + lx * prob(y0, x1)) + ly * (hx * prob(y1, x0) + lx * prob(y1, x1));
        v = val > thresh ? 1 : 0;
    }
    out[p] = v;
}

// ---- connected components (4-connectivity, scipy.ndimage.label's default structure) by union-find on pixel indices -------------------
__device__ inline int uf_find(const int* lab, int i) {
    int r = i;
    while (true) { const int p = lab[r]; if (p == r) break; r = p; }
    return r;
}
__device__ inline void uf_union(int* lab, int a, int b) {
    while (true) {
        a = uf_find(lab, a); b = uf_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }               // a > b: hang a under b
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;                                                    // somebody else re-parented a: retry from there
    }
}
// grid = (ceil(HW/256), R, B); lab / cnt: [B*R][HW] int32.  Slots that cannot become planes (beyond n_det, or score <= confidence:
// select_top_predictions drops them before overlay_mask looks at their masks) are skipped by every kernel.
__device__ inline bool slot_live(const float* det_scores, const int32_t* n_det, int b, int k, int R, float confidence) {
    return k < n_det[b] && det_scores[(size_t)b * R + k] > confidence;
}
__global__ void __launch_bounds__(256)
ccl_init_kernel(const uint8_t* __restrict__ m, int* __restrict__ lab, int* __restrict__ cnt, int* __restrict__ biggest, int* __restrict__ n_ties, int HW,
                int W, const float* __restrict__ det_scores, const int32_t* __restrict__ n_det, float confidence) {
    const int b = blockIdx.z, k = blockIdx.y, R = gridDim.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) { biggest[b * R + k] = 0; n_ties[b * R + k] = 0; }
    if (!slot_live(det_scores, n_det, b, k, R, confidence)) return;                 // uniform per workgroup
    const size_t base = ((size_t)b * R + k) * HW;
    // Every pixel starts out pointing at the first pixel of its horizontal run inside this wave's 64 pixels (two ballots), so the
    // merge pass only has to join runs across wave boundaries and rows -- not every pair of neighbours.
    const int lane = threadIdx.x & 63;
    const bool on = p < HW && m[base + p];
    const unsigned long long bal = __ballot(on);
    const bool prev = lane > 0 && ((bal >> (lane - 1)) & 1ull);
    const bool start = on && (!prev || (p % W) == 0);
    const unsigned long long sb = __ballot(start);
    if (p >= HW) return;
    int l = -1;
    if (on) l = p - (lane - (63 - __clzll((long long)(sb & ((2ull << lane) - 1ull)))));
    lab[base + p] = l;
    cnt[base + p] = 0;
}
__global__ void __launch_bounds__(256)
ccl_merge_kernel(const uint8_t* __restrict__ m, int* __restrict__ lab, int H, int W, const float* __restrict__ det_scores,
                 const int32_t* __restrict__ n_det, float confidence) {
    const int b = blockIdx.z, k = blockIdx.y, R = gridDim.y;
    const size_t base = ((size_t)b * R + k) * H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W || !slot_live(det_scores, n_det, b, k, R, confidence) || !m[base + p]) return;
    const int y = p / W, x = p - y * W;
    const bool left = x > 0 && m[base + p - 1];
    if ((threadIdx.x & 63) == 0 && left) uf_union(lab + base, p, p - 1);              // a run that continues across the wave boundary
    // rows: one union per overlapping segment (its first pixel), not one per pixel
    if (y + 1 < H && m[base + p + W] && !(left && m[base + p + W - 1])) uf_union(lab + base, p, p + W);
}
// Flattens the labels and counts the component sizes.  The lanes of a wave that found the same root add their number with ONE atomic
// (neighbouring pixels mostly share a component; a device-scope atomic per pixel on a handful of addresses serialises: 330 us per
// image before).  biggest[slot] = running maximum of the totals the atomics return = the size of the biggest component at the end.
__global__ void __launch_bounds__(256)
ccl_count_kernel(int* __restrict__ lab, int* __restrict__ cnt, int* __restrict__ biggest, int HW, const float* __restrict__ det_scores,
                 const int32_t* __restrict__ n_det, float confidence) {
    const int b = blockIdx.z, k = blockIdx.y, R = gridDim.y;
    const size_t base = ((size_t)b * R + k) * HW;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (!slot_live(det_scores, n_det, b, k, R, confidence)) return;
    int r = -1;
    if (p < HW && lab[base + p] >= 0) {
        r = uf_find(lab + base, p);
        lab[base + p] = r;                                          // (roots keep pointing at themselves: concurrent finds stay valid)
    }
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(r >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lr = __shfl(r, leader, 64);
        const unsigned long long same = __ballot(r == lr) & todo;
        if (lane == leader) {
            const int add = __popcll(same);
            const int total = atomicAdd(&cnt[base + lr], add) + add;
            atomicMax(&biggest[b * R + k], total);
        }
        todo &= ~same;
    }
}
// number of components of the maximal size per slot (get_biggest_plane keeps ALL of them, predictor.py:317, and overlay_mask sizes
// the plane by their sum): roots whose count equals the maximum
__global__ void __launch_bounds__(256)
ccl_ties_kernel(const int* __restrict__ lab, const int* __restrict__ cnt, const int* __restrict__ biggest, int* __restrict__ n_ties, int HW,
                const float* __restrict__ det_scores, const int32_t* __restrict__ n_det, float confidence) {
    const int b = blockIdx.z, k = blockIdx.y, R = gridDim.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW || !slot_live(det_scores, n_det, b, k, R, confidence)) return;
    const size_t base = ((size_t)b * R + k) * HW;
    if (lab[base + p] == p && cnt[base + p] == biggest[b * R + k]) atomicAdd(&n_ties[b * R + k], 1);
}

// One workgroup (64 threads) per image: confident detections by descending score (select_top_predictions), then the stable sort by
// biggest-component size (sorted(..., reverse=True) keeps the score order among equal sizes), area filter; rank[k] = id of slot k or 0.
__global__ void __launch_bounds__(64)
plane_rank_kernel(const float* __restrict__ det_scores, const int32_t* __restrict__ n_det, const int* __restrict__ biggest,
                  const int* __restrict__ n_ties, int R, float confidence, float min_size,
                  int* __restrict__ order /*[B][R]: slot painted i-th, -1 = none*/) {
    __shared__ float key[64];
    __shared__ int idx[64];
    __shared__ int by_score[64];
    const int b = blockIdx.x, t = threadIdx.x;
    const bool ok = t < R && t < n_det[b] && det_scores[(size_t)b * R + t] > confidence;
    key[t] = ok ? det_scores[(size_t)b * R + t] : -1.f;
    idx[t] = t;
    bitonic_sort_desc(key, idx, 64);                                   // ties: lower slot first (torch.sort: unspecified)
    by_score[t] = key[t] > 0.f ? idx[t] : -1;
    __syncthreads();
    const int slot = by_score[t];
    const int size = slot >= 0 ? biggest[b * R + slot] * max(n_ties[b * R + slot], 1) : -1;
    key[t] = (slot >= 0 && (float)size >= min_size) ? (float)size : -1.f;       // sizes <= 76800 are exact in fp32
    idx[t] = t;                                                         // index = rank by score: ties keep the score order
    bitonic_sort_desc(key, idx, 64);
    if (t < R) order[(size_t)b * R + t] = key[t] > 0.f ? by_score[idx[t]] : -1;
}
// inst[b][p] = 1 + (last i such that pixel p belongs to the biggest component(s) of slot order[i]); painted in order, later overwrites
__global__ void __launch_bounds__(256)
plane_paint_kernel(const int* __restrict__ lab, const int* __restrict__ cnt, const int* __restrict__ biggest, const int* __restrict__ order,
                   int R, int HW, uint8_t* __restrict__ inst) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    int id = 0;
    for (int i = 0; i < R; ++i) {
        const int slot = order[(size_t)b * R + i];
        if (slot < 0) break;
        const size_t base = ((size_t)b * R + slot) * HW;
        const int r = lab[base + p];
        if (r >= 0 && cnt[base + r] == biggest[b * R + slot]) id = i + 1;      // every component of the maximal size (predictor.py:317)
    }
    inst[(size_t)b * HW + p] = (uint8_t)id;
}

}  // namespace

extern "C" int vidc_det_stem_im2col(const float* image01_nchw, float* cols, int B, int H, int W, int Hp, int Wp, float mean_b, float mean_g,
                                    float mean_r, vidc_stream_t stream) {
    VIDC_REQUIRE(image01_nchw && cols, VIDC_ERR_NULL, "vidc_det_stem_im2col: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Hp % 2 == 0 && Wp % 2 == 0, VIDC_ERR_SHAPE, "vidc_det_stem_im2col: bad shape");
    const long long total = (long long)B * (Hp / 2) * (Wp / 2) * 54;
    hipLaunchKernelGGL(det_stem_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), image01_nchw, cols, B,
                       H, W, Hp, Wp, mean_b, mean_g, mean_r);
    VIDC_CHECK_LAUNCH("det_stem_im2col_kernel");
    return VIDC_OK;
}

extern "C" int vidc_upsample_nearest2x(const float* x, float* y, int B, int h, int w, int C, int ldx, int ldy, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y, VIDC_ERR_NULL, "vidc_upsample_nearest2x: null pointer");
    VIDC_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, VIDC_ERR_SHAPE, "vidc_upsample_nearest2x: bad shape");
    const long long total = (long long)B * 4 * h * w * (C / 4);
    hipLaunchKernelGGL(upsample_nearest2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vidc::as_stream(stream), x, y, B, h, w, C, ldx, ldy);
    VIDC_CHECK_LAUNCH("upsample_nearest2x_kernel");
    return VIDC_OK;
}

extern "C" int vidc_rpn_topk_decode(const float* rpn_map, int B, int h, int w, int ld, int A, int stride, const float* cell_anchors_host,
                                    int pre_nms_top_n, int img_h, int img_w, float* boxes, float* scores, long long out_stride,
                                    vidc_stream_t stream) {
    VIDC_REQUIRE(rpn_map && cell_anchors_host && boxes && scores, VIDC_ERR_NULL, "vidc_rpn_topk_decode: null pointer");
    const int n = A * h * w;
    VIDC_REQUIRE(B > 0 && A == 3 && n > 0 && n <= 16384 && ld >= 5 * A, VIDC_ERR_SHAPE, "vidc_rpn_topk_decode: needs A == 3 and A*h*w <= 16384 (got %d)", n);
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int k = pre_nms_top_n < n ? pre_nms_top_n : n;
    Anchors cell;
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 4; ++c) cell.a[a][c] = cell_anchors_host[a * 4 + c];
    const size_t lds = (size_t)n2 * 8;
    static bool attr_done = false;
    if (!attr_done) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_topk_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_topk_select_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (16384 + 2048 + 2048) * 4));
        attr_done = true;
    }
    if (n > 1024 && k <= 1024) {
        hipLaunchKernelGGL(rpn_topk_select_decode_kernel, dim3(B), dim3(1024), (size_t)(n + 2048 + 2048) * 4, vidc::as_stream(stream), rpn_map, h, w,
                           ld, A, stride, cell, k, (float)img_w, (float)img_h, boxes, scores, out_stride);
        VIDC_CHECK_LAUNCH("rpn_topk_select_decode_kernel");
        return VIDC_OK;
    }
    hipLaunchKernelGGL(rpn_topk_decode_kernel, dim3(B), dim3(1024), lds, vidc::as_stream(stream), rpn_map, h, w, ld, A, stride, cell, n2, k,
                       (float)img_w, (float)img_h, boxes, scores, out_stride);
    VIDC_CHECK_LAUNCH("rpn_topk_decode_kernel");
    return VIDC_OK;
}

extern "C" int vidc_rpn_topk_decode_levels(const float* const* maps_host, const int32_t* hw_host, const int32_t* strides_host,
                                           const float* cell_anchors_host, const int32_t* level_offsets_host, int n_levels, int B, int ld, int A,
                                           int pre_nms_top_n, int img_h, int img_w, float* boxes, float* scores, long long out_stride,
                                           vidc_stream_t stream) {
    VIDC_REQUIRE(maps_host && hw_host && strides_host && cell_anchors_host && level_offsets_host && boxes && scores, VIDC_ERR_NULL,
                 "vidc_rpn_topk_decode_levels: null pointer");
    VIDC_REQUIRE(B > 0 && A == 3 && n_levels >= 1 && n_levels <= 5 && ld >= 5 * A, VIDC_ERR_SHAPE, "vidc_rpn_topk_decode_levels: needs A == 3, <= 5 levels");
    LevelSet ls;
    size_t lds = 0;
    for (int l = 0; l < n_levels; ++l) {
        LevelDesc& d = ls.lv[l];
        d.map = maps_host[l]; d.h = hw_host[2 * l]; d.w = hw_host[2 * l + 1]; d.stride = strides_host[l]; d.off = level_offsets_host[l];
        VIDC_REQUIRE(d.map, VIDC_ERR_NULL, "vidc_rpn_topk_decode_levels: null map");
        const int n = A * d.h * d.w;
        VIDC_REQUIRE(n > 0 && n <= 16384, VIDC_ERR_SHAPE, "vidc_rpn_topk_decode_levels: A*h*w = %d out of range (1..16384)", n);
        d.k = pre_nms_top_n < n ? pre_nms_top_n : n;
        d.n2 = 64;
        while (d.n2 < n) d.n2 <<= 1;
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 4; ++c) d.cell.a[a][c] = cell_anchors_host[(l * 3 + a) * 4 + c];
        const size_t need = (n > 1024 && d.k <= 1024) ? (size_t)(n + 2048 + 2048) * 4 : (size_t)d.n2 * 8;
        lds = need > lds ? need : lds;
    }
    static bool attr_done = false;
    if (!attr_done) {
        VIDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_topk_levels_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
        attr_done = true;
    }
    VIDC_REQUIRE(lds <= 16384 * 8, VIDC_ERR_SHAPE, "vidc_rpn_topk_decode_levels: LDS need %zu", lds);
    hipLaunchKernelGGL(rpn_topk_levels_kernel, dim3(B, n_levels), dim3(1024), lds, vidc::as_stream(stream), ls, ld, A, (float)img_w, (float)img_h,
                       boxes, scores, out_stride);
    VIDC_CHECK_LAUNCH("rpn_topk_levels_kernel");
    return VIDC_OK;
}

extern "C" int vidc_rpn_select(const float* boxes, const float* scores, const int32_t* keep, const int32_t* n_keep, int B, int n_levels,
                               const int32_t* level_offsets_host, int per_level, int total, float* proposals, float* proposal_scores,
                               int32_t* n_proposals, vidc_stream_t stream) {
    VIDC_REQUIRE(boxes && scores && keep && n_keep && level_offsets_host && proposals && proposal_scores && n_proposals, VIDC_ERR_NULL,
                 "vidc_rpn_select: null pointer");
    VIDC_REQUIRE(B > 0 && n_levels > 0 && n_levels <= 7 && per_level > 0 && total > 0 && n_levels * per_level <= 512, VIDC_ERR_SHAPE,
                 "vidc_rpn_select: at most 512 candidates per image");
    Levels lv;
    lv.n_levels = n_levels;
    for (int l = 0; l <= n_levels; ++l) lv.off[l] = level_offsets_host[l];
    hipLaunchKernelGGL(rpn_select_kernel, dim3(B), dim3(256), 0, vidc::as_stream(stream), boxes, scores, keep, n_keep, lv, lv.off[n_levels],
                       per_level, total, proposals, proposal_scores, n_proposals);
    VIDC_CHECK_LAUNCH("rpn_select_kernel");
    return VIDC_OK;
}

extern "C" int vidc_roi_align_fpn(const float* const* feats_host, const int32_t* hw_host, int n_levels, int C, const float* boxes, int B, int R,
                                  int pooled, int sampling_ratio, float* y, vidc_stream_t stream) {
    VIDC_REQUIRE(feats_host && hw_host && boxes && y, VIDC_ERR_NULL, "vidc_roi_align_fpn: null pointer");
    VIDC_REQUIRE(n_levels == 4 && C > 0 && B > 0 && R > 0 && pooled > 0, VIDC_ERR_SHAPE, "vidc_roi_align_fpn: needs the 4 levels P2..P5");
    Pyramid py;
    for (int l = 0; l < 4; ++l) {
        py.x[l] = feats_host[l]; py.h[l] = hw_host[2 * l]; py.w[l] = hw_host[2 * l + 1];
        py.scale[l] = 1.0f / (float)(4 << l);
        VIDC_REQUIRE(py.x[l], VIDC_ERR_NULL, "vidc_roi_align_fpn: null feature map");
    }
    const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    hipLaunchKernelGGL(roi_align_fpn_kernel, dim3(pooled * pooled, B * R), dim3(threads), 0, vidc::as_stream(stream), py, C, boxes, R, pooled,
                       sampling_ratio, y);
    VIDC_CHECK_LAUNCH("roi_align_fpn_kernel");
    return VIDC_OK;
}

extern "C" int vidc_det_candidates(const float* head_out, int ld, const float* proposals, const int32_t* n_proposals, int B, int R, int img_h,
                                   int img_w, float score_thresh, float* cand_boxes, float* cand_scores, int32_t* cand_src, int32_t* n_cand,
                                   vidc_stream_t stream) {
    VIDC_REQUIRE(head_out && proposals && n_proposals && cand_boxes && cand_scores && cand_src && n_cand, VIDC_ERR_NULL, "vidc_det_candidates: null pointer");
    VIDC_REQUIRE(B > 0 && R > 0 && R <= 64 && ld >= 10, VIDC_ERR_SHAPE, "vidc_det_candidates: at most 64 proposals per image");
    hipLaunchKernelGGL(det_candidates_kernel, dim3(B), dim3(64), 0, vidc::as_stream(stream), head_out, ld, proposals, n_proposals, R, (float)img_w,
                       (float)img_h, score_thresh, cand_boxes, cand_scores, cand_src, n_cand);
    VIDC_CHECK_LAUNCH("det_candidates_kernel");
    return VIDC_OK;
}

extern "C" int vidc_det_select(const float* cand_boxes, const float* cand_scores, const int32_t* cand_src, const int32_t* n_cand,
                               const int32_t* keep, const int32_t* n_keep, int B, int R, float* det_boxes, float* det_scores, int32_t* n_det,
                               vidc_stream_t stream) {
    VIDC_REQUIRE(cand_boxes && cand_scores && cand_src && n_cand && keep && n_keep && det_boxes && det_scores && n_det, VIDC_ERR_NULL,
                 "vidc_det_select: null pointer");
    VIDC_REQUIRE(B > 0 && R > 0 && R <= 64, VIDC_ERR_SHAPE, "vidc_det_select: at most 64 slots per image");
    hipLaunchKernelGGL(det_select_kernel, dim3(B), dim3(64), 0, vidc::as_stream(stream), cand_boxes, cand_scores, cand_src, n_cand, keep, n_keep, R,
                       det_boxes, det_scores, n_det);
    VIDC_CHECK_LAUNCH("det_select_kernel");
    return VIDC_OK;
}

extern "C" int vidc_mask_paste(const float* mask_logits, int ld, int cls, int M, const float* det_boxes, const int32_t* n_det, int B, int R, int H,
                               int W, float thresh, uint8_t* pasted, vidc_stream_t stream) {
    VIDC_REQUIRE(mask_logits && det_boxes && n_det && pasted, VIDC_ERR_NULL, "vidc_mask_paste: null pointer");
    VIDC_REQUIRE(B > 0 && R > 0 && H > 0 && W > 0 && M > 0 && M % 2 == 0 && cls >= 0 && cls < ld, VIDC_ERR_SHAPE, "vidc_mask_paste: bad shape");
    hipLaunchKernelGGL(mask_paste_kernel, dim3(vidc::cdiv(H * W, 256), R, B), dim3(256), 0, vidc::as_stream(stream), mask_logits, ld, cls, M, det_boxes,
                       n_det, R, H, W, thresh, pasted);
    VIDC_CHECK_LAUNCH("mask_paste_kernel");
    return VIDC_OK;
}

extern "C" size_t vidc_instance_map_scratch_bytes(int B, int R, int H, int W) {
    return ((size_t)B * R * H * W * 2 + (size_t)B * R * 3 + 64) * sizeof(int32_t);
}

extern "C" int vidc_instance_map(const uint8_t* pasted, const float* det_scores, const int32_t* n_det, int B, int R, int H, int W, float confidence,
                                 float min_fraction, uint8_t* inst, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(pasted && det_scores && n_det && inst && scratch, VIDC_ERR_NULL, "vidc_instance_map: null pointer");
    VIDC_REQUIRE(B > 0 && R > 0 && R <= 64 && H > 0 && W > 0 && (long long)H * W < (1 << 24), VIDC_ERR_SHAPE, "vidc_instance_map: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int HW = H * W;
    int* lab = reinterpret_cast<int*>(scratch);
    int* cnt = lab + (size_t)B * R * HW;
    int* biggest = cnt + (size_t)B * R * HW;
    int* order = biggest + (size_t)B * R;
    int* n_ties = order + (size_t)B * R;
    const dim3 grid(vidc::cdiv(HW, 256), R, B);
    hipLaunchKernelGGL(ccl_init_kernel, grid, dim3(256), 0, st, pasted, lab, cnt, biggest, n_ties, HW, W, det_scores, n_det, confidence);
    VIDC_CHECK_LAUNCH("ccl_init_kernel");
    hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(256), 0, st, pasted, lab, H, W, det_scores, n_det, confidence);
    VIDC_CHECK_LAUNCH("ccl_merge_kernel");
    hipLaunchKernelGGL(ccl_count_kernel, grid, dim3(256), 0, st, lab, cnt, biggest, HW, det_scores, n_det, confidence);
    VIDC_CHECK_LAUNCH("ccl_count_kernel");
    hipLaunchKernelGGL(ccl_ties_kernel, grid, dim3(256), 0, st, lab, cnt, biggest, n_ties, HW, det_scores, n_det, confidence);
    VIDC_CHECK_LAUNCH("ccl_ties_kernel");
    hipLaunchKernelGGL(plane_rank_kernel, dim3(B), dim3(64), 0, st, det_scores, n_det, biggest, n_ties, R, confidence, min_fraction * (float)HW,
                       order);
    VIDC_CHECK_LAUNCH("plane_rank_kernel");
    hipLaunchKernelGGL(plane_paint_kernel, dim3(vidc::cdiv(HW, 256), B), dim3(256), 0, st, lab, cnt, biggest, order, R, HW, inst);
    VIDC_CHECK_LAUNCH("plane_paint_kernel");
    return VIDC_OK;
}
