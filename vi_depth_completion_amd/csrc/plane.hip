// Plane block on device: RANSAC mean normal, plane-offset RANSAC, normal->depth plane projection, sparse-depth
// override + candidate counts, and enrichment scatter.  Replaces main.py:29-190 and :285-294 of the reference, where
// each plane costs ~470 ATen calls and dozens of host syncs; here every stage is a launch over ALL plane "slots"
// (slot = one plane id of one image of the batch) x pixel chunks, reductions are two-level with a fixed order
// (bit-reproducible), and the host is synchronised exactly once per batch (the read of the enrichment counts).
//
// Random draws stay on the host (numpy legacy RNG, exactly like the reference) and arrive as index arrays.
#include "common.h"

namespace {

constexpr int CH = 256;                   // pixels per workgroup chunk
constexpr float ANGLE_THR = 20.0f;        // MEAN_NORMAL_ANGLE_DIFF_THR / angle_threshold_degrees (main.py:25,38)
constexpr float RAD2DEG = (float)(180.0 / 3.14159265358979323846);
constexpr float COS_THR = 0.93969262078590838f;   // cos(20 deg)
constexpr float DIST_THR = 1.0e-1f;       // plane_offset_ransac distance_threshold (main.py:68)
constexpr float MAX_DEPTH_MULT = 10.f;    // main.py:22
constexpr float MAX_DEPTH = 10.f;         // main.py:23

struct Slot { int b, cls, hyp_off, n_hyp; };

__device__ inline bool close_angle(float dot) {
    // torch: acos(clamp(dot,-1,1)) * (180/pi) < 20.  Far from the threshold the comparison is decided on the cosine;
    // only borderline values pay for the acosf.
    dot = fminf(fmaxf(dot, -1.0f), 1.0f);
    if (dot > COS_THR + 1e-4f) return true;
    if (dot < COS_THR - 1e-4f) return false;
    return acosf(dot) * RAD2DEG < ANGLE_THR;
}

__device__ inline float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

// Sum over the 256 threads of a workgroup; fixed order -> deterministic.  Result valid in every thread.
template <typename T>
__device__ inline T block_sum(T v, T* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    T s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// torch.argmax (first maximal index) over n >= 0 non-negative counts, computed by ONE WHOLE WAVE (all 64 lanes must call it):
// lane l scans l, l+64, ...; ties resolve to the smaller index in the lane scan and in the shuffle tree.  Same value in every lane.
__device__ inline int first_argmax(const int32_t* c, int n) {
    const int lane = threadIdx.x & 63;
    int best = 0x7fffffff, bc = -1;
    for (int h = lane; h < n; h += 64) { const int v = c[h]; if (v > bc) { bc = v; best = h; } }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int oc = __shfl_xor(bc, off, 64), ob = __shfl_xor(best, off, 64);
        if (oc > bc || (oc == bc && ob < best)) { bc = oc; best = ob; }
    }
    return n > 0 ? best : 0;
}

// ---- stage 1a: inlier counts of every hypothesis.  threads = hypotheses; 256 pixels of one slot are compacted into LDS (ballot
//      ranks, one float4 per pixel, zero-padded to a multiple of 4), the inner loop takes 4 points per step with the four broadcast
//      ds_read_b128 issued up front and the rare borderline acosf kept out of the straight-line code -----------------------------
__global__ void __launch_bounds__(320)
ransac_count_kernel(const float* __restrict__ normals, const uint8_t* __restrict__ ids, const Slot* __restrict__ slots,
                    const int32_t* __restrict__ hyp_pix, int HW, int32_t* __restrict__ counts) {
    __shared__ float4 pts[CH + 4];
    __shared__ int wcnt[CH / 64];
    const Slot s = slots[blockIdx.y];
    const float* nb = normals + (size_t)s.b * 3 * HW;
    const uint8_t* idb = ids + (size_t)s.b * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * CH + threadIdx.x;
    const bool mine = threadIdx.x < CH && p < HW && idb[p] == (uint8_t)s.cls;
    const unsigned long long bal = __ballot(mine);
    if (lane == 0 && wave < CH / 64) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int n = 0, k = __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < CH / 64; ++w) { n += wcnt[w]; if (w < wave) k += wcnt[w]; }
    if (n == 0) return;
    if (mine) pts[k] = make_float4(nb[p], nb[HW + p], nb[2 * HW + p], 0.f);
    if (threadIdx.x < 4) pts[n + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);     // dot = 0: neither close nor borderline
    __syncthreads();
    if ((int)threadIdx.x >= s.n_hyp) return;
    const int hp = hyp_pix[s.hyp_off + threadIdx.x];
    const float hx = nb[hp], hy = nb[HW + hp], hz = nb[2 * HW + hp];
    int c = 0;
    for (int i = 0; i < n; i += 4) {
        const float4 q0 = pts[i], q1 = pts[i + 1], q2 = pts[i + 2], q3 = pts[i + 3];
        float t[4] = {dot3(hx, hy, hz, q0.x, q0.y, q0.z), dot3(hx, hy, hz, q1.x, q1.y, q1.z), dot3(hx, hy, hz, q2.x, q2.y, q2.z),
                      dot3(hx, hy, hz, q3.x, q3.y, q3.z)};
        bool edge = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {               // same decision as close_angle(): clamp, cosine far from the threshold, acosf near it
            t[j] = fminf(fmaxf(t[j], -1.0f), 1.0f);
            c += t[j] > COS_THR + 1e-4f ? 1 : 0;
            edge |= fabsf(t[j] - COS_THR) <= 1e-4f;
        }
        if (__builtin_expect(edge, 0)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fabsf(t[j] - COS_THR) <= 1e-4f) c += acosf(t[j]) * RAD2DEG < ANGLE_THR ? 1 : 0;
        }
    }
    if (c) atomicAdd(&counts[blockIdx.y * VIDC_MAX_HYP + threadIdx.x], c);     // integer: order-independent
}

// ---- stage 1b: inlier mask of the best hypothesis + per-chunk partial sums of the inlier normals ---------------------
__global__ void __launch_bounds__(CH)
ransac_mask_kernel(const float* __restrict__ normals, const uint8_t* __restrict__ ids, const Slot* __restrict__ slots,
                   const int32_t* __restrict__ hyp_pix, int HW, const int32_t* __restrict__ counts,
                   uint8_t* __restrict__ inlier_mask, float* __restrict__ partial /* [slot][chunk][4] */) {
    __shared__ float redf[CH / 64];
    __shared__ int s_best;
    const Slot s = slots[blockIdx.y];
    const float* nb = normals + (size_t)s.b * 3 * HW;
    if (threadIdx.x < 64) {
        const int bst = first_argmax(counts + blockIdx.y * VIDC_MAX_HYP, s.n_hyp);
        if (threadIdx.x == 0) s_best = bst;
    }
    __syncthreads();
    const int bp = s.n_hyp > 0 ? hyp_pix[s.hyp_off + s_best] : 0;
    const float hx = nb[bp], hy = nb[HW + bp], hz = nb[2 * HW + bp];
    const int p = blockIdx.x * CH + threadIdx.x;
    bool in = false;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (p < HW && s.n_hyp > 0 && ids[(size_t)s.b * HW + p] == (uint8_t)s.cls) {
        nx = nb[p]; ny = nb[HW + p]; nz = nb[2 * HW + p];
        in = close_angle(dot3(hx, hy, hz, nx, ny, nz));
    }
    if (p < HW) inlier_mask[(size_t)blockIdx.y * HW + p] = in ? 1 : 0;
    const float sx = block_sum(in ? nx : 0.f, redf), sy = block_sum(in ? ny : 0.f, redf), sz = block_sum(in ? nz : 0.f, redf);
    const float cn = block_sum(in ? 1.f : 0.f, redf);
    if (threadIdx.x == 0) {
        float* q = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        q[0] = sx; q[1] = sy; q[2] = sz; q[3] = cn;
    }
}

// mean_normal (main.py:29-34) from the chunk partials; fixed two-level order (thread t owns chunks t, t+256, ...), so every
// workgroup that recomputes it gets the same bits.
__device__ inline void mean_normal_from_partials(const float* partial, int n_chunks, float* redf /* LDS [CH/64] */, float& mx,
                                                 float& my, float& mz, int& n_in) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int c = threadIdx.x; c < n_chunks; c += CH) {
        const float4 q = *reinterpret_cast<const float4*>(partial + (size_t)c * 4);
        s0 += q.x; s1 += q.y; s2 += q.z; s3 += q.w;
    }
    s0 = block_sum(s0, redf); s1 = block_sum(s1, redf); s2 = block_sum(s2, redf); s3 = block_sum(s3, redf);
    n_in = (int)s3;
    mx = my = mz = 0.f;
    if (n_in > 0) {
        mx = s0 / (float)n_in; my = s1 / (float)n_in; mz = s2 / (float)n_in;
        const float nrm = fmaxf(sqrtf(mx * mx + my * my + mz * mz), 1e-12f);   // F.normalize
        mx /= nrm; my /= nrm; mz /= nrm;
    }
}

// ---- stage 1c: per-chunk partial sums of |angle(n, n_bar)| over the inliers ---------------------------------------------
__global__ void __launch_bounds__(CH)
ransac_angle_kernel(const float* __restrict__ normals, const Slot* __restrict__ slots, int HW,
                    const uint8_t* __restrict__ inlier_mask, const float* __restrict__ partial, float* __restrict__ apartial) {
    __shared__ float redf[CH / 64];
    const Slot s = slots[blockIdx.y];
    const float* nb = normals + (size_t)s.b * 3 * HW;
    float mx, my, mz; int n_in;
    mean_normal_from_partials(partial + (size_t)blockIdx.y * gridDim.x * 4, gridDim.x, redf, mx, my, mz, n_in);
    const int p = blockIdx.x * CH + threadIdx.x;
    float a = 0.f;
    if (p < HW && inlier_mask[(size_t)blockIdx.y * HW + p]) {
        const float d = fminf(fmaxf(dot3(nb[p], nb[HW + p], nb[2 * HW + p], mx, my, mz), -1.f), 1.f);
        a = fabsf(acosf(d) * RAD2DEG);
    }
    a = block_sum(a, redf);
    if (threadIdx.x == 0) apartial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = a;
}

// ---- stage 2: record + plane offset from the (few) sparse depths on the inliers.  One workgroup per slot. ---------------
// sparse_idx: per image, the flat indices of the pixels with depth > 0 in row-major order (boolean-indexing order),
// built once per batch by sparse_list_kernel; n_sparse[b] their count.
__global__ void __launch_bounds__(CH)
plane_offset_kernel(const float* __restrict__ homo, const float* __restrict__ depth, const Slot* __restrict__ slots,
                    const uint8_t* __restrict__ inlier_mask, int HW, int n_chunks, const float* __restrict__ partial,
                    const float* __restrict__ apartial, const int32_t* __restrict__ sparse_idx, const int32_t* __restrict__ n_sparse,
                    int max_sparse, const int32_t* __restrict__ counts, float* __restrict__ records,
                    const int32_t* __restrict__ dense_hyp, const int32_t* __restrict__ dense_n, float* __restrict__ dense_dots) {
    __shared__ float redf[CH / 64];
    __shared__ int redi[CH / 64];
    __shared__ float dots[VIDC_MAX_HYP];
    __shared__ int hcnt[VIDC_MAX_HYP];
    __shared__ int s_n, s_best;
    __shared__ float s_dsum;
    const Slot s = slots[blockIdx.x];
    float* rec = records + (size_t)blockIdx.x * VIDC_PLANE_RECORD;
    float mx, my, mz; int n_in;
    mean_normal_from_partials(partial + (size_t)blockIdx.x * n_chunks * 4, n_chunks, redf, mx, my, mz, n_in);
    float asum = 0.f;
    for (int c = threadIdx.x; c < n_chunks; c += CH) asum += apartial[(size_t)blockIdx.x * n_chunks + c];
    // fixed-order second level: thread t owns chunks t, t+256, ...; block_sum is order-fixed too
    asum = block_sum(asum, redf);
    const float mean_angle = n_in > 0 ? asum / (float)n_in : 0.f;
    const bool accepted = n_in > 0 && !(mean_angle > ANGLE_THR);            // main.py:162
    if (threadIdx.x < 64) {
        const int bst = first_argmax(counts + blockIdx.x * VIDC_MAX_HYP, s.n_hyp);
        if (threadIdx.x == 0) {
            rec[0] = mx; rec[1] = my; rec[2] = mz; rec[3] = 0.f; rec[4] = (float)n_in; rec[5] = mean_angle; rec[6] = accepted ? 1.f : 0.f;
            for (int i = 7; i < VIDC_PLANE_RECORD; ++i) rec[i] = 0.f;
            rec[12] = (float)bst;
            s_n = 0; s_dsum = 0.f;
        }
    }
    __syncthreads();
    if (!accepted) return;
    // ordered compaction of the sparse points that lie on the plane (the list is already in row-major order)
    const float* hb = homo + (size_t)s.b * HW * 3;
    const float* db = depth + (size_t)s.b * HW;
    const uint8_t* mk = inlier_mask + (size_t)blockIdx.x * HW;
    const int ns = min(n_sparse[s.b], max_sparse);
    {
        __shared__ int wcnt[CH / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int total = 0;
        float dloc = 0.f;
        for (int base = 0; base < ns; base += CH) {       // uniform trip count; ranks keep the list's row-major order
            const int i = base + threadIdx.x;
            bool f = false;
            float d = 0.f, dt = 0.f;
            if (i < ns) {
                const int p = sparse_idx[(size_t)s.b * max_sparse + i];
                if (mk[p]) {
                    f = true;
                    d = db[p];
                    dt = dot3(mx, my, mz, hb[p * 3] * d, hb[p * 3 + 1] * d, hb[p * 3 + 2] * d);
                }
            }
            const unsigned long long m = __ballot(f);
            __syncthreads();
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            int k = total;
            for (int w = 0; w < wave; ++w) k += wcnt[w];
            k += __popcll(m & ((1ull << lane) - 1ull));
            if (f && k < VIDC_MAX_HYP) dots[k] = dt;
            if (f && dense_dots) dense_dots[(size_t)blockIdx.x * HW + k] = dt;      // every on-plane point, for the subsampled branch
            dloc += d;
            for (int w = 0; w < CH / 64; ++w) total += wcnt[w];
        }
        dloc = block_sum(dloc, redf);
        if (threadIdx.x == 0) { s_n = total; s_dsum = dloc; }
    }
    __syncthreads();
    const int n_pts = s_n;
    float offset = 0.f;
    int n_inl = 0;
    if (n_pts > VIDC_MAX_HYP) {
        // main.py:75-78: more points than hypotheses -> the hypotheses are np.random.permutation(np.r_[0:n_pts])[0:300], a HOST draw
        // (numpy's legacy generator, in the reference's draw order).  First pass: flag the slot and report n_pts; the host draws
        // and calls again with dense_hyp[slot] = the 300 ranks and dense_n[slot] = the n_pts they were drawn for.
        if (!dense_hyp || !dense_n || !dense_dots || dense_n[blockIdx.x] != n_pts) {
            if (threadIdx.x == 0) { rec[7] = (float)n_pts; rec[9] = 0.f; rec[10] = -1.f; rec[6] = 0.f; }
            return;
        }
        __threadfence_block();
        __syncthreads();
        const float* dd = dense_dots + (size_t)blockIdx.x * HW;
        const int32_t* hy = dense_hyp + (size_t)blockIdx.x * VIDC_MAX_HYP;
        for (int j = threadIdx.x; j < VIDC_MAX_HYP; j += CH) {
            int c = 0;
            const float hyp = -dd[hy[j]];
            for (int i = 0; i < n_pts; ++i) c += (fabsf(hyp + dd[i]) < DIST_THR) ? 1 : 0;      // dd[i]: one broadcast load per step
            hcnt[j] = c;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int bst = first_argmax(hcnt, VIDC_MAX_HYP);
            if (threadIdx.x == 0) s_best = bst;
        }
        __syncthreads();
        const float hyp = -dd[hy[s_best]];
        float sdot = 0.f; int c = 0;
        for (int i = threadIdx.x; i < n_pts; i += CH)
            if (fabsf(hyp + dd[i]) < DIST_THR) { sdot += dd[i]; ++c; }
        sdot = block_sum(sdot, redf);
        n_inl = block_sum(c, redi);
        offset = n_inl > 0 ? -(sdot / (float)n_inl) : 0.f;
    } else if (n_pts == 1) {
        offset = -dots[0]; n_inl = 1;               // main.py:81-83
    } else if (n_pts > 1) {
        for (int j = threadIdx.x; j < n_pts; j += CH) {
            int c = 0;
            const float hyp = -dots[j];
            for (int i = 0; i < n_pts; ++i) c += (fabsf(hyp + dots[i]) < DIST_THR) ? 1 : 0;
            hcnt[j] = c;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int bst = first_argmax(hcnt, n_pts);
            if (threadIdx.x == 0) s_best = bst;
        }
        __syncthreads();
        const float hyp = -dots[s_best];
        float sdot = 0.f; int c = 0;
        for (int i = threadIdx.x; i < n_pts; i += CH)
            if (fabsf(hyp + dots[i]) < DIST_THR) { sdot += dots[i]; ++c; }
        sdot = block_sum(sdot, redf);
        n_inl = block_sum(c, redi);
        offset = n_inl > 0 ? -(sdot / (float)n_inl) : 0.f;
    }
    if (threadIdx.x == 0) {
        rec[3] = offset;
        rec[7] = (float)n_pts;
        rec[8] = n_pts > 0 ? s_dsum / (float)n_pts : 0.f;
        rec[9] = (float)n_inl;
    }
}

// Row-major list of the pixels with depth > 0 of every image.  One workgroup (16 waves) per image; wave w owns the w-th contiguous
// sixteenth of the map (a multiple of 256 pixels).  Pass 1 counts the wave's candidates (independent float4 loads, no barrier),
// one barrier publishes the 16 totals, pass 2 re-reads the segment (cache hits) and emits the indices in row-major order: inside a
// 256-pixel step the order is lane-major / component-minor, so a pixel's rank is four masked popcounts plus its lane-local prefix.
__global__ void __launch_bounds__(1024)
sparse_list_kernel(const float* __restrict__ depth, int HW, int max_sparse, int32_t* __restrict__ sparse_idx,
                   int32_t* __restrict__ n_sparse) {
    __shared__ int wcount[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* db = depth + (size_t)b * HW;
    const int seg = ((HW + 16 * 256 - 1) / (16 * 256)) * 256;
    const int lo = wv * seg, hi = min(lo + seg, HW);
    const bool vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(db) & 15) == 0;
    auto load4 = [&](int p, bool (&f)[4]) {
        if (vec && p + 3 < hi) {
            const float4 v = *reinterpret_cast<const float4*>(db + p);
            f[0] = v.x > 0.f; f[1] = v.y > 0.f; f[2] = v.z > 0.f; f[3] = v.w > 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) f[c] = p + c < hi && db[p + c] > 0.f;
        }
    };
    int cnt = 0;
    int p0 = lo;
    if (vec) {                                         // straight-line body: 4 independent 16-byte loads in flight per lane
        for (; p0 + 1024 <= hi; p0 += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(db + p0 + u * 256 + lane * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) cnt += (int)(v[u].x > 0.f) + (int)(v[u].y > 0.f) + (int)(v[u].z > 0.f) + (int)(v[u].w > 0.f);
        }
    }
    for (; p0 < hi; p0 += 256) {
        bool f[4];
        load4(p0 + lane * 4, f);
        cnt += (int)f[0] + (int)f[1] + (int)f[2] + (int)f[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) wcount[wv] = cnt;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const int c = wcount[w]; total += c; if (w < wv) base += c; }
    if (threadIdx.x == 0) n_sparse[b] = total;
    const unsigned long long below = (1ull << lane) - 1ull;
    int32_t* out = sparse_idx + (size_t)b * max_sparse;
    auto emit = [&](int p, const bool (&f)[4]) {       // 256 pixels: p = this lane's first one
        const unsigned long long b0 = __ballot(f[0]), b1 = __ballot(f[1]), b2 = __ballot(f[2]), b3 = __ballot(f[3]);
        if ((b0 | b1 | b2 | b3) == 0ull) return;
        int k = base + __popcll(b0 & below) + __popcll(b1 & below) + __popcll(b2 & below) + __popcll(b3 & below);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (f[c]) { if (k < max_sparse) out[k] = p + c; ++k; }
        base += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
    };
    p0 = lo;
    if (vec) {
        for (; p0 + 1024 <= hi; p0 += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(db + p0 + u * 256 + lane * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool f[4] = {v[u].x > 0.f, v[u].y > 0.f, v[u].z > 0.f, v[u].w > 0.f};
                emit(p0 + u * 256 + lane * 4, f);
            }
        }
    }
    for (; p0 < hi; p0 += 256) {
        bool f[4];
        load4(p0 + lane * 4, f);
        emit(p0 + lane * 4, f);
    }
}

// ---- stage 3: depth = -d / (n . homo) on the plane: per-chunk validity statistics, then the masked write ---------------
__device__ inline bool plane_value(const float* hb, int p, float nx, float ny, float nz, float d, float& v) {
    const float dots = (hb[p * 3] * nx + hb[p * 3 + 1] * ny) + hb[p * 3 + 2] * nz;
    if (!(fabsf(dots) > 1e-3f)) return false;
    v = -d / dots;
    return true;
}

__global__ void __launch_bounds__(CH)
plane_stats_kernel(const float* __restrict__ homo, const Slot* __restrict__ slots, const uint8_t* __restrict__ inlier_mask, int HW,
                   const float* __restrict__ records, int32_t* __restrict__ stats /* [slot][chunk][4] */) {
    __shared__ int redi[CH / 64];
    const Slot s = slots[blockIdx.y];
    const float* rec = records + (size_t)blockIdx.y * VIDC_PLANE_RECORD;
    int32_t* q = stats + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    if (rec[6] == 0.f || rec[9] == 0.f) {            // rejected plane, or no offset inliers (main.py:176-178)
        if (threadIdx.x < 4) q[threadIdx.x] = 0;
        return;
    }
    const float* hb = homo + (size_t)s.b * HW * 3;
    const int p = blockIdx.x * CH + threadIdx.x;
    int n = 0, n_big = 0, n_over = 0, n_neg = 0;
    float v;
    if (p < HW && inlier_mask[(size_t)blockIdx.y * HW + p] && plane_value(hb, p, rec[0], rec[1], rec[2], rec[3], v)) {
        n = 1; n_big = v > rec[8] * MAX_DEPTH_MULT; n_over = v > MAX_DEPTH; n_neg = v < 0.f;
    }
    n = block_sum(n, redi); n_big = block_sum(n_big, redi); n_over = block_sum(n_over, redi); n_neg = block_sum(n_neg, redi);
    if (threadIdx.x == 0) { q[0] = n; q[1] = n_big; q[2] = n_over; q[3] = n_neg; }
}

__global__ void __launch_bounds__(CH)
plane_write_kernel(const float* __restrict__ homo, const Slot* __restrict__ slots, const uint8_t* __restrict__ inlier_mask, int HW,
                   float* __restrict__ records, const int32_t* __restrict__ stats, float* __restrict__ plane_depth) {
    __shared__ int redi[CH / 64];
    const Slot s = slots[blockIdx.y];
    float* rec = records + (size_t)blockIdx.y * VIDC_PLANE_RECORD;
    if (rec[6] == 0.f || rec[9] == 0.f) return;
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;              // integer sums: any order gives the same totals
    for (int c = threadIdx.x; c < (int)gridDim.x; c += CH) {
        const int32_t* q = stats + ((size_t)blockIdx.y * gridDim.x + c) * 4;
        t0 += q[0]; t1 += q[1]; t2 += q[2]; t3 += q[3];
    }
    const int n = block_sum(t0, redi), n_big = block_sum(t1, redi), n_over = block_sum(t2, redi), n_neg = block_sum(t3, redi);
    bool valid = true;                                 // main.py:120-125 (true division)
    if (n > 0) {
        if ((float)n_big / (float)n > 0.05f || n_over > 0) valid = false;
        if (n_neg > 0) valid = false;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { rec[10] = valid ? 1.f : 0.f; rec[11] = (float)n; }
    if (!valid) return;
    const float* hb = homo + (size_t)s.b * HW * 3;
    const int p = blockIdx.x * CH + threadIdx.x;
    float v;
    if (p < HW && inlier_mask[(size_t)blockIdx.y * HW + p] && plane_value(hb, p, rec[0], rec[1], rec[2], rec[3], v))
        plane_depth[(size_t)s.b * HW + p] = v;
}

// ---- stage 4: sparse depths override the plane depths; per-chunk counts of the enrichment candidates ---------------------
// info layout (int32): [B][n_chunks] chunk counts of (plane_depth > 0), then [1] = number of slots flagged -1.
__global__ void __launch_bounds__(CH)
plane_finalize_kernel(const float* __restrict__ depth, float* __restrict__ plane_depth, int HW, const float* __restrict__ records,
                      int n_slots, int32_t* __restrict__ info) {
    __shared__ int redi[CH / 64];
    const int b = blockIdx.y;
    const int p = blockIdx.x * CH + threadIdx.x;
    int nz = 0;
    if (p < HW) {
        const float d = depth[(size_t)b * HW + p];
        float v = plane_depth[(size_t)b * HW + p];
        if (d > 0.f) { v = d; plane_depth[(size_t)b * HW + p] = d; }
        nz = v > 0.f;
    }
    nz = block_sum(nz, redi);
    if (threadIdx.x == 0) info[(size_t)b * gridDim.x + blockIdx.x] = nz;
    if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
        int flagged = 0;
        for (int s = 0; s < n_slots; ++s) flagged += records[(size_t)s * VIDC_PLANE_RECORD + 10] < 0.f;
        info[(size_t)gridDim.y * gridDim.x] = flagged;
    }
}

// ---- stage 5: copy the sub[]-th nonzeros (row-major order) of plane_depth into the enriched sparse depth -----------
// chunk_base: [B][n_chunks] exclusive prefix of the chunk counts (computed by the host from `info`).
// `sparse` != NULL: enriched = clone(sparse) with the selected plane depths copied in, in ONE pass (main.py:286 + :293-294) --
// every pixel is written, so `enriched` needs no initialisation; NULL: only the selected pixels are written (the caller cloned).
__global__ void __launch_bounds__(CH)
enrich_scatter_kernel(const float* __restrict__ plane_depth, const float* __restrict__ sparse, const int32_t* __restrict__ sub,
                      const int32_t* __restrict__ sub_off, const int32_t* __restrict__ chunk_base, int HW, float* __restrict__ enriched) {
    __shared__ int wave_base[CH / 64];
    const int b = blockIdx.y;
    const int s0 = sub_off[b], n_sub = sub_off[b + 1] - s0;
    const int p = blockIdx.x * CH + threadIdx.x;
    if (n_sub <= 0) {
        if (sparse && p < HW) enriched[(size_t)b * HW + p] = sparse[(size_t)b * HW + p];
        return;
    }
    const float v = p < HW ? plane_depth[(size_t)b * HW + p] : 0.f;
    const bool nz = v > 0.f;
    const unsigned long long m = __ballot(nz);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_base[wave] = __popcll(m);
    __syncthreads();
    int k = chunk_base[(size_t)b * gridDim.x + blockIdx.x];
    for (int w = 0; w < wave; ++w) k += wave_base[w];
    k += __popcll(m & ((1ull << lane) - 1ull));       // row-major rank of this pixel among the nonzeros of image b
    bool hit = false;
    if (nz) {
        int lo = 0, hi = n_sub;                        // sub is sorted and unique: binary search for k
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sub[s0 + mid] < k) lo = mid + 1; else hi = mid; }
        hit = lo < n_sub && sub[s0 + lo] == k;
    }
    if (hit) enriched[(size_t)b * HW + p] = v;
    else if (sparse && p < HW) enriched[(size_t)b * HW + p] = sparse[(size_t)b * HW + p];
}

}  // namespace

extern "C" size_t vidc_plane_scratch_bytes(int n_slots, int B, int HW) {
    const size_t nc = (size_t)vidc::cdiv(HW, CH);
    // partial [slots][nc][4] f32 + apartial [slots][nc] f32 + stats [slots][nc][4] i32 + sparse list [B][HW] + n_sparse[B]
    // (the list holds EVERY pixel with depth > 0: dense depth inputs take the subsampled branch of plane_offset_ransac, main.py:75-78)
    return (size_t)n_slots * nc * (4 + 1 + 4) * 4 + (size_t)B * ((size_t)HW + 1) * 4 + 256;
}

extern "C" int vidc_plane_ransac_normal(const float* normals, const uint8_t* ids, const int32_t* slots, int n_slots,
                                        const int32_t* hyp_pix, int HW, uint8_t* inlier_mask, int32_t* counts, void* scratch,
                                        vidc_stream_t stream) {
    VIDC_REQUIRE(normals && ids && slots && hyp_pix && inlier_mask && counts && scratch, VIDC_ERR_NULL,
                 "vidc_plane_ransac_normal: null pointer");
    VIDC_REQUIRE(n_slots > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_plane_ransac_normal: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int nc = vidc::cdiv(HW, CH);
    float* partial = reinterpret_cast<float*>(scratch);
    float* apartial = partial + (size_t)n_slots * nc * 4;
    VIDC_HIP(hipMemsetAsync(counts, 0, (size_t)n_slots * VIDC_MAX_HYP * sizeof(int32_t), st));
    const Slot* sl = reinterpret_cast<const Slot*>(slots);
    hipLaunchKernelGGL(ransac_count_kernel, dim3(nc, n_slots), dim3(320), 0, st, normals, ids, sl, hyp_pix, HW, counts);
    VIDC_CHECK_LAUNCH("ransac_count_kernel");
    hipLaunchKernelGGL(ransac_mask_kernel, dim3(nc, n_slots), dim3(CH), 0, st, normals, ids, sl, hyp_pix, HW, counts, inlier_mask, partial);
    VIDC_CHECK_LAUNCH("ransac_mask_kernel");
    hipLaunchKernelGGL(ransac_angle_kernel, dim3(nc, n_slots), dim3(CH), 0, st, normals, sl, HW, inlier_mask, partial, apartial);
    VIDC_CHECK_LAUNCH("ransac_angle_kernel");
    return VIDC_OK;
}

extern "C" int vidc_plane_offset(const float* homo, const float* depth, const int32_t* slots, int n_slots, int B,
                                 const uint8_t* inlier_mask, const int32_t* counts, int HW, void* scratch, float* records,
                                 vidc_stream_t stream) {
    return vidc_plane_offset_dense(homo, depth, slots, n_slots, B, inlier_mask, counts, HW, scratch, records, nullptr, nullptr, nullptr, stream);
}

extern "C" int vidc_plane_offset_dense(const float* homo, const float* depth, const int32_t* slots, int n_slots, int B,
                                       const uint8_t* inlier_mask, const int32_t* counts, int HW, void* scratch, float* records,
                                       const int32_t* dense_hyp, const int32_t* dense_n, float* dense_dots, vidc_stream_t stream) {
    VIDC_REQUIRE(homo && depth && slots && inlier_mask && counts && scratch && records, VIDC_ERR_NULL, "vidc_plane_offset: null pointer");
    VIDC_REQUIRE((!dense_hyp && !dense_n && !dense_dots) || (dense_hyp && dense_n && dense_dots), VIDC_ERR_NULL,
                 "vidc_plane_offset_dense: dense_hyp, dense_n and dense_dots go together");
    VIDC_REQUIRE(n_slots > 0 && HW > 0 && B > 0, VIDC_ERR_SHAPE, "vidc_plane_offset: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int nc = vidc::cdiv(HW, CH);
    float* partial = reinterpret_cast<float*>(scratch);
    float* apartial = partial + (size_t)n_slots * nc * 4;
    int32_t* sparse_idx = reinterpret_cast<int32_t*>(apartial + (size_t)n_slots * nc) + (size_t)n_slots * nc * 4;
    int32_t* n_sparse = sparse_idx + (size_t)B * HW;
    hipLaunchKernelGGL(sparse_list_kernel, dim3(B), dim3(1024), 0, st, depth, HW, HW, sparse_idx, n_sparse);
    VIDC_CHECK_LAUNCH("sparse_list_kernel");
    hipLaunchKernelGGL(plane_offset_kernel, dim3(n_slots), dim3(CH), 0, st, homo, depth, reinterpret_cast<const Slot*>(slots),
                       inlier_mask, HW, nc, partial, apartial, sparse_idx, n_sparse, HW, counts, records, dense_hyp, dense_n, dense_dots);
    VIDC_CHECK_LAUNCH("plane_offset_kernel");
    return VIDC_OK;
}

extern "C" int vidc_plane_project_depth(const float* homo, const int32_t* slots, int n_slots, const uint8_t* inlier_mask, int HW,
                                        void* scratch, float* records, float* plane_depth, vidc_stream_t stream) {
    VIDC_REQUIRE(homo && slots && inlier_mask && scratch && records && plane_depth, VIDC_ERR_NULL, "vidc_plane_project_depth: null pointer");
    VIDC_REQUIRE(n_slots > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_plane_project_depth: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int nc = vidc::cdiv(HW, CH);
    int32_t* stats = reinterpret_cast<int32_t*>(reinterpret_cast<float*>(scratch) + (size_t)n_slots * nc * 5);
    const Slot* sl = reinterpret_cast<const Slot*>(slots);
    hipLaunchKernelGGL(plane_stats_kernel, dim3(nc, n_slots), dim3(CH), 0, st, homo, sl, inlier_mask, HW, records, stats);
    VIDC_CHECK_LAUNCH("plane_stats_kernel");
    hipLaunchKernelGGL(plane_write_kernel, dim3(nc, n_slots), dim3(CH), 0, st, homo, sl, inlier_mask, HW, records, stats, plane_depth);
    VIDC_CHECK_LAUNCH("plane_write_kernel");
    return VIDC_OK;
}

extern "C" int vidc_plane_info_count(int B, int HW) { return B * vidc::cdiv(HW, CH) + 1; }

extern "C" int vidc_plane_finalize(const float* depth, float* plane_depth, int B, int HW, const float* records, int n_slots,
                                   int32_t* info, vidc_stream_t stream) {
    VIDC_REQUIRE(depth && plane_depth && info && (records || n_slots == 0), VIDC_ERR_NULL, "vidc_plane_finalize: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0 && n_slots >= 0, VIDC_ERR_SHAPE, "vidc_plane_finalize: bad shape");
    hipLaunchKernelGGL(plane_finalize_kernel, dim3(vidc::cdiv(HW, CH), B), dim3(CH), 0, vidc::as_stream(stream), depth, plane_depth, HW,
                       records, n_slots, info);
    VIDC_CHECK_LAUNCH("plane_finalize_kernel");
    return VIDC_OK;
}

// The plane block of one batch as ONE call: plane_depth <- depth (device copy), vidc_plane_ransac_normal, vidc_plane_offset[_dense],
// vidc_plane_project_depth, vidc_plane_finalize, in that order on `stream` -- the same kernels with the same arguments as the five
// separate entries (bit-identical results); what it saves is four trips through the caller's language per item of a frame stream, during
// which the lane's stream has nothing queued (round 4: the stream modes enqueue these launch-bound kernels between two graph segments).
// n_slots == 0 (only background): copy + finalize.  dense_* all NULL: the ordinary offset pass.
extern "C" int vidc_plane_block(const float* normals, const uint8_t* ids, const int32_t* slots, int n_slots, const int32_t* hyp_pix, int B, int HW,
                                const float* homo, const float* depth, uint8_t* inlier_mask, int32_t* counts, void* scratch, float* records,
                                const int32_t* dense_hyp, const int32_t* dense_n, float* dense_dots, float* plane_depth, int32_t* info,
                                vidc_stream_t stream) {
    VIDC_REQUIRE(depth && plane_depth && info, VIDC_ERR_NULL, "vidc_plane_block: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0 && n_slots >= 0, VIDC_ERR_SHAPE, "vidc_plane_block: bad shape");
    VIDC_HIP(hipMemcpyAsync(plane_depth, depth, (size_t)B * HW * sizeof(float), hipMemcpyDeviceToDevice, vidc::as_stream(stream)));
    if (n_slots > 0) {
        int rc = vidc_plane_ransac_normal(normals, ids, slots, n_slots, hyp_pix, HW, inlier_mask, counts, scratch, stream);
        if (rc != VIDC_OK) return rc;
        if (dense_hyp || dense_n || dense_dots)
            rc = vidc_plane_offset_dense(homo, depth, slots, n_slots, B, inlier_mask, counts, HW, scratch, records, dense_hyp, dense_n, dense_dots, stream);
        else
            rc = vidc_plane_offset(homo, depth, slots, n_slots, B, inlier_mask, counts, HW, scratch, records, stream);
        if (rc != VIDC_OK) return rc;
        rc = vidc_plane_project_depth(homo, slots, n_slots, inlier_mask, HW, scratch, records, plane_depth, stream);
        if (rc != VIDC_OK) return rc;
    }
    return vidc_plane_finalize(depth, plane_depth, B, HW, records, n_slots, info, stream);
}

extern "C" int vidc_enrich_scatter(const float* plane_depth, const int32_t* sub, const int32_t* sub_offsets, const int32_t* chunk_base,
                                   int B, int HW, float* enriched, vidc_stream_t stream) {
    VIDC_REQUIRE(plane_depth && sub && sub_offsets && chunk_base && enriched, VIDC_ERR_NULL, "vidc_enrich_scatter: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_enrich_scatter: bad shape");
    hipLaunchKernelGGL(enrich_scatter_kernel, dim3(vidc::cdiv(HW, CH), B), dim3(CH), 0, vidc::as_stream(stream), plane_depth,
                       (const float*)nullptr, sub, sub_offsets, chunk_base, HW, enriched);
    VIDC_CHECK_LAUNCH("enrich_scatter_kernel");
    return VIDC_OK;
}

extern "C" int vidc_enrich_scatter_from(const float* plane_depth, const float* sparse_depth, const int32_t* sub, const int32_t* sub_offsets,
                                        const int32_t* chunk_base, int B, int HW, float* enriched, vidc_stream_t stream) {
    VIDC_REQUIRE(plane_depth && sparse_depth && sub && sub_offsets && chunk_base && enriched, VIDC_ERR_NULL, "vidc_enrich_scatter_from: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0 && enriched != sparse_depth, VIDC_ERR_SHAPE, "vidc_enrich_scatter_from: bad shape (enriched must not alias sparse_depth)");
    hipLaunchKernelGGL(enrich_scatter_kernel, dim3(vidc::cdiv(HW, CH), B), dim3(CH), 0, vidc::as_stream(stream), plane_depth, sparse_depth,
                       sub, sub_offsets, chunk_base, HW, enriched);
    VIDC_CHECK_LAUNCH("enrich_scatter_kernel");
    return VIDC_OK;
}
