// Plane block on device: RANSAC mean normal, plane-offset RANSAC, normal->depth plane projection, sparse-depth
// override + count, and enrichment scatter.  Replaces main.py:29-190 and :285-294 of the reference, where each plane
// costs ~470 ATen calls and dozens of host syncs; here every stage is ONE launch over all plane "slots"
// (slot = one plane id of one image of the batch) and nothing synchronises with the host.
//
// Random draws stay on the host (numpy legacy RNG, exactly like the reference) and arrive as index arrays.
#include "common.h"

namespace {

constexpr int NT = 1024;                  // threads of the single-workgroup-per-slot kernels
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

template <typename T>
__device__ inline T block_sum(T v, T* red) {          // all NT threads call; result broadcast
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    T s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];   // fixed order -> deterministic
    return s;
}

// ---- stage 1a: inlier counts of every hypothesis (grid: pixel chunks x slots) -----------------------------------
__global__ void __launch_bounds__(256)
ransac_count_kernel(const float* __restrict__ normals, const uint8_t* __restrict__ ids, const Slot* __restrict__ slots,
                    const int32_t* __restrict__ hyp_pix, int HW, int32_t* __restrict__ counts) {
    __shared__ float hn[VIDC_MAX_HYP * 3];
    __shared__ int cnt[VIDC_MAX_HYP];
    const Slot s = slots[blockIdx.y];
    const float* nb = normals + (size_t)s.b * 3 * HW;
    const uint8_t* idb = ids + (size_t)s.b * HW;
    for (int h = threadIdx.x; h < s.n_hyp; h += blockDim.x) {
        int p = hyp_pix[s.hyp_off + h];
        hn[h * 3 + 0] = nb[p]; hn[h * 3 + 1] = nb[HW + p]; hn[h * 3 + 2] = nb[2 * HW + p];
        cnt[h] = 0;
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = p < HW && idb[p] == (uint8_t)s.cls;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (mine) { nx = nb[p]; ny = nb[HW + p]; nz = nb[2 * HW + p]; }
    const int lane = threadIdx.x & 63;
    if (__ballot(mine) != 0ull) {
        for (int h = 0; h < s.n_hyp; ++h) {
            bool c = mine && close_angle(dot3(hn[h * 3], hn[h * 3 + 1], hn[h * 3 + 2], nx, ny, nz));
            unsigned long long m = __ballot(c);
            if (lane == 0 && m) atomicAdd(&cnt[h], __popcll(m));
        }
    }
    __syncthreads();
    for (int h = threadIdx.x; h < s.n_hyp; h += blockDim.x)
        if (cnt[h]) atomicAdd(&counts[blockIdx.y * VIDC_MAX_HYP + h], cnt[h]);
}

// ---- stage 1b: best hypothesis -> inlier mask, mean normal, mean angle (one workgroup per slot) ----------------------
__global__ void __launch_bounds__(NT)
ransac_select_kernel(const float* __restrict__ normals, const uint8_t* __restrict__ ids, const Slot* __restrict__ slots,
                     const int32_t* __restrict__ hyp_pix, int HW, const int32_t* __restrict__ counts,
                     uint8_t* __restrict__ inlier_mask, float* __restrict__ records) {
    __shared__ float redf[NT / 64];
    __shared__ int redi[NT / 64];
    __shared__ int s_best;
    const Slot s = slots[blockIdx.x];
    const float* nb = normals + (size_t)s.b * 3 * HW;
    const uint8_t* idb = ids + (size_t)s.b * HW;
    uint8_t* mk = inlier_mask + (size_t)blockIdx.x * HW;
    float* rec = records + (size_t)blockIdx.x * VIDC_PLANE_RECORD;
    if (threadIdx.x == 0) {          // torch.argmax: first maximal index
        int best = 0, bc = -1;
        for (int h = 0; h < s.n_hyp; ++h) {
            int c = counts[blockIdx.x * VIDC_MAX_HYP + h];
            if (c > bc) { bc = c; best = h; }
        }
        s_best = best;
    }
    __syncthreads();
    const int bp = s.n_hyp > 0 ? hyp_pix[s.hyp_off + s_best] : 0;
    const float hx = nb[bp], hy = nb[HW + bp], hz = nb[2 * HW + bp];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int n_in = 0;
    const int per = vidc::cdiv(HW, NT);
    const int p0 = threadIdx.x * per, p1 = min(HW, p0 + per);
    for (int p = p0; p < p1; ++p) {
        bool in = false;
        if (s.n_hyp > 0 && idb[p] == (uint8_t)s.cls) {
            float nx = nb[p], ny = nb[HW + p], nz = nb[2 * HW + p];
            in = close_angle(dot3(hx, hy, hz, nx, ny, nz));
            if (in) { sx += nx; sy += ny; sz += nz; ++n_in; }
        }
        mk[p] = in ? 1 : 0;
    }
    sx = block_sum(sx, redf); sy = block_sum(sy, redf); sz = block_sum(sz, redf);
    n_in = block_sum(n_in, redi);
    // mean_normal (main.py:29-34): F.normalize(mean)
    float mx = 0.f, my = 0.f, mz = 0.f;
    if (n_in > 0) {
        mx = sx / (float)n_in; my = sy / (float)n_in; mz = sz / (float)n_in;
        float nrm = fmaxf(sqrtf(mx * mx + my * my + mz * mz), 1e-12f);
        mx /= nrm; my /= nrm; mz /= nrm;
    }
    float asum = 0.f;
    for (int p = p0; p < p1; ++p)
        if (mk[p]) {
            float d = fminf(fmaxf(dot3(nb[p], nb[HW + p], nb[2 * HW + p], mx, my, mz), -1.f), 1.f);
            asum += fabsf(acosf(d) * RAD2DEG);
        }
    asum = block_sum(asum, redf);
    if (threadIdx.x == 0) {
        float mean_angle = n_in > 0 ? asum / (float)n_in : 0.f;
        rec[0] = mx; rec[1] = my; rec[2] = mz; rec[3] = 0.f;
        rec[4] = (float)n_in; rec[5] = mean_angle;
        rec[6] = (n_in > 0 && !(mean_angle > ANGLE_THR)) ? 1.f : 0.f;
        for (int i = 7; i < VIDC_PLANE_RECORD; ++i) rec[i] = 0.f;
        rec[12] = (float)s_best;
    }
}

// ---- stage 2: plane offset from the sparse depths on the inliers (one workgroup per slot) ---------------------------
__global__ void __launch_bounds__(NT)
plane_offset_kernel(const float* __restrict__ homo, const float* __restrict__ depth, const Slot* __restrict__ slots,
                    const uint8_t* __restrict__ inlier_mask, int HW, float* __restrict__ records) {
    __shared__ float dots[VIDC_MAX_HYP];
    __shared__ int redi[NT / 64];
    __shared__ float redf[NT / 64];
    __shared__ int scan[NT];
    __shared__ int hcnt[VIDC_MAX_HYP];
    __shared__ int s_best;
    const Slot s = slots[blockIdx.x];
    float* rec = records + (size_t)blockIdx.x * VIDC_PLANE_RECORD;
    if (rec[6] == 0.f) return;                      // plane rejected by the normal test (main.py:162)
    const float* hb = homo + (size_t)s.b * HW * 3;
    const float* db = depth + (size_t)s.b * HW;
    const uint8_t* mk = inlier_mask + (size_t)blockIdx.x * HW;
    const float nx = rec[0], ny = rec[1], nz = rec[2];
    const int per = vidc::cdiv(HW, NT);
    const int p0 = threadIdx.x * per, p1 = min(HW, p0 + per);
    int mine = 0;
    float dsum = 0.f;
    for (int p = p0; p < p1; ++p)
        if (mk[p] && db[p] > 0.f) { ++mine; dsum += db[p]; }
    // ordered compaction (row-major like boolean indexing): exclusive scan of per-thread counts
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    const int n_pts = scan[NT - 1];
    int k = scan[threadIdx.x] - mine;
    dsum = block_sum(dsum, redf);
    if (n_pts > VIDC_MAX_HYP) {                      // would need the host permutation (main.py:78): flagged, not faked
        if (threadIdx.x == 0) { rec[7] = (float)n_pts; rec[9] = 0.f; rec[10] = -1.f; rec[6] = 0.f; }
        return;
    }
    for (int p = p0; p < p1; ++p)
        if (mk[p] && db[p] > 0.f) {
            float d = db[p];
            dots[k++] = dot3(nx, ny, nz, hb[p * 3] * d, hb[p * 3 + 1] * d, hb[p * 3 + 2] * d);
        }
    __syncthreads();
    float offset = 0.f;
    int n_inl = 0;
    if (n_pts == 1) {
        offset = -dots[0]; n_inl = 1;               // main.py:81-83
    } else if (n_pts > 1) {
        for (int j = threadIdx.x; j < n_pts; j += NT) {
            int c = 0;
            const float hyp = -dots[j];
            for (int i = 0; i < n_pts; ++i) c += (fabsf(hyp + dots[i]) < DIST_THR) ? 1 : 0;
            hcnt[j] = c;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int best = 0, bc = -1;
            for (int j = 0; j < n_pts; ++j) if (hcnt[j] > bc) { bc = hcnt[j]; best = j; }
            s_best = best;
        }
        __syncthreads();
        const float hyp = -dots[s_best];
        float sdot = 0.f; int c = 0;
        for (int i = threadIdx.x; i < n_pts; i += NT)
            if (fabsf(hyp + dots[i]) < DIST_THR) { sdot += dots[i]; ++c; }
        sdot = block_sum(sdot, redf);
        n_inl = block_sum(c, redi);
        offset = n_inl > 0 ? -(sdot / (float)n_inl) : 0.f;
    }
    if (threadIdx.x == 0) {
        rec[3] = offset;
        rec[7] = (float)n_pts;
        rec[8] = n_pts > 0 ? dsum / (float)n_pts : 0.f;
        rec[9] = (float)n_inl;
    }
}

// ---- stage 3: depth = -d / (n . homo) on the plane, validity tests, masked write (one workgroup per slot) ------------
__global__ void __launch_bounds__(NT)
plane_project_kernel(const float* __restrict__ homo, const Slot* __restrict__ slots, const uint8_t* __restrict__ inlier_mask,
                     int HW, float* __restrict__ records, float* __restrict__ plane_depth) {
    __shared__ int redi[NT / 64];
    const Slot s = slots[blockIdx.x];
    float* rec = records + (size_t)blockIdx.x * VIDC_PLANE_RECORD;
    if (rec[6] == 0.f || rec[9] == 0.f) return;      // rejected plane, or no offset inliers (main.py:176-178)
    const float* hb = homo + (size_t)s.b * HW * 3;
    const uint8_t* mk = inlier_mask + (size_t)blockIdx.x * HW;
    float* out = plane_depth + (size_t)s.b * HW;
    const float nx = rec[0], ny = rec[1], nz = rec[2], d = rec[3], mean_depth = rec[8];
    const int per = vidc::cdiv(HW, NT);
    const int p0 = threadIdx.x * per, p1 = min(HW, p0 + per);
    int n = 0, n_big = 0, n_over = 0, n_neg = 0;
    for (int p = p0; p < p1; ++p) {
        if (!mk[p]) continue;
        float dots = (hb[p * 3] * nx + hb[p * 3 + 1] * ny) + hb[p * 3 + 2] * nz;
        if (!(fabsf(dots) > 1e-3f)) continue;
        float v = -d / dots;
        ++n;
        n_big += v > mean_depth * MAX_DEPTH_MULT;
        n_over += v > MAX_DEPTH;
        n_neg += v < 0.f;
    }
    n = block_sum(n, redi); n_big = block_sum(n_big, redi); n_over = block_sum(n_over, redi); n_neg = block_sum(n_neg, redi);
    bool valid = true;
    if (n > 0) {
        if ((float)n_big / (float)n > 0.05f || n_over > 0) valid = false;
        if (n_neg > 0) valid = false;
    }
    if (threadIdx.x == 0) { rec[10] = valid ? 1.f : 0.f; rec[11] = (float)n; }
    if (!valid) return;
    for (int p = p0; p < p1; ++p) {
        if (!mk[p]) continue;
        float dots = (hb[p * 3] * nx + hb[p * 3 + 1] * ny) + hb[p * 3 + 2] * nz;
        if (fabsf(dots) > 1e-3f) out[p] = -d / dots;
    }
}

// ---- stage 4: sparse depths override the plane depths; count the candidates for enrichment ---------------------------
__global__ void __launch_bounds__(256)
plane_finalize_kernel(const float* __restrict__ depth, float* __restrict__ plane_depth, int HW, int32_t* __restrict__ nnz) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool nz = false;
    if (p < HW) {
        float d = depth[(size_t)b * HW + p];
        float v = plane_depth[(size_t)b * HW + p];
        if (d > 0.f) { v = d; plane_depth[(size_t)b * HW + p] = d; }
        nz = v > 0.f;
    }
    unsigned long long m = __ballot(nz);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&nnz[b], __popcll(m));
}

// ---- stage 5: copy the sub[]-th nonzeros (row-major order) of plane_depth into the enriched sparse depth -----------
__global__ void __launch_bounds__(NT)
enrich_scatter_kernel(const float* __restrict__ plane_depth, const int32_t* __restrict__ sub, const int32_t* __restrict__ sub_off,
                      int HW, float* __restrict__ enriched) {
    __shared__ int scan[NT];
    const int b = blockIdx.x;
    const float* pd = plane_depth + (size_t)b * HW;
    float* en = enriched + (size_t)b * HW;
    const int s0 = sub_off[b], n_sub = sub_off[b + 1] - s0;
    if (n_sub <= 0) return;
    const int per = vidc::cdiv(HW, NT);
    const int p0 = threadIdx.x * per, p1 = min(HW, p0 + per);
    int mine = 0;
    for (int p = p0; p < p1; ++p) mine += pd[p] > 0.f;
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    int k = scan[threadIdx.x] - mine;
    // first entry of sub[] that is >= k (sub is sorted, unique)
    int lo = 0, hi = n_sub;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (sub[s0 + mid] < k) lo = mid + 1; else hi = mid; }
    for (int p = p0; p < p1 && lo < n_sub; ++p)
        if (pd[p] > 0.f) {
            if (sub[s0 + lo] == k) { en[p] = pd[p]; ++lo; }
            ++k;
        }
}

}  // namespace

extern "C" int vidc_plane_ransac_normal(const float* normals, const uint8_t* ids, const int32_t* slots, int n_slots,
                                        const int32_t* hyp_pix, int HW, uint8_t* inlier_mask, int32_t* counts, float* records,
                                        vidc_stream_t stream) {
    VIDC_REQUIRE(normals && ids && slots && hyp_pix && inlier_mask && counts && records, VIDC_ERR_NULL,
                 "vidc_plane_ransac_normal: null pointer");
    VIDC_REQUIRE(n_slots > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_plane_ransac_normal: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    VIDC_HIP(hipMemsetAsync(counts, 0, (size_t)n_slots * VIDC_MAX_HYP * sizeof(int32_t), st));
    const Slot* sl = reinterpret_cast<const Slot*>(slots);
    hipLaunchKernelGGL(ransac_count_kernel, dim3(vidc::cdiv(HW, 256), n_slots), dim3(256), 0, st, normals, ids, sl, hyp_pix, HW, counts);
    VIDC_CHECK_LAUNCH("ransac_count_kernel");
    hipLaunchKernelGGL(ransac_select_kernel, dim3(n_slots), dim3(NT), 0, st, normals, ids, sl, hyp_pix, HW, counts, inlier_mask, records);
    VIDC_CHECK_LAUNCH("ransac_select_kernel");
    return VIDC_OK;
}

extern "C" int vidc_plane_offset(const float* homo, const float* depth, const int32_t* slots, int n_slots,
                                 const uint8_t* inlier_mask, int HW, float* records, vidc_stream_t stream) {
    VIDC_REQUIRE(homo && depth && slots && inlier_mask && records, VIDC_ERR_NULL, "vidc_plane_offset: null pointer");
    VIDC_REQUIRE(n_slots > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_plane_offset: bad shape");
    hipLaunchKernelGGL(plane_offset_kernel, dim3(n_slots), dim3(NT), 0, vidc::as_stream(stream), homo, depth,
                       reinterpret_cast<const Slot*>(slots), inlier_mask, HW, records);
    VIDC_CHECK_LAUNCH("plane_offset_kernel");
    return VIDC_OK;
}

extern "C" int vidc_plane_project_depth(const float* homo, const int32_t* slots, int n_slots, const uint8_t* inlier_mask, int HW,
                                        float* records, float* plane_depth, vidc_stream_t stream) {
    VIDC_REQUIRE(homo && slots && inlier_mask && records && plane_depth, VIDC_ERR_NULL, "vidc_plane_project_depth: null pointer");
    VIDC_REQUIRE(n_slots > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_plane_project_depth: bad shape");
    hipLaunchKernelGGL(plane_project_kernel, dim3(n_slots), dim3(NT), 0, vidc::as_stream(stream), homo,
                       reinterpret_cast<const Slot*>(slots), inlier_mask, HW, records, plane_depth);
    VIDC_CHECK_LAUNCH("plane_project_kernel");
    return VIDC_OK;
}

extern "C" int vidc_plane_finalize(const float* depth, float* plane_depth, int B, int HW, int32_t* nnz_out, vidc_stream_t stream) {
    VIDC_REQUIRE(depth && plane_depth && nnz_out, VIDC_ERR_NULL, "vidc_plane_finalize: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_plane_finalize: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    VIDC_HIP(hipMemsetAsync(nnz_out, 0, (size_t)B * sizeof(int32_t), st));
    hipLaunchKernelGGL(plane_finalize_kernel, dim3(vidc::cdiv(HW, 256), B), dim3(256), 0, st, depth, plane_depth, HW, nnz_out);
    VIDC_CHECK_LAUNCH("plane_finalize_kernel");
    return VIDC_OK;
}

extern "C" int vidc_enrich_scatter(const float* plane_depth, const int32_t* sub, const int32_t* sub_offsets, int B, int HW,
                                   float* enriched, vidc_stream_t stream) {
    VIDC_REQUIRE(plane_depth && sub && sub_offsets && enriched, VIDC_ERR_NULL, "vidc_enrich_scatter: null pointer");
    VIDC_REQUIRE(B > 0 && HW > 0, VIDC_ERR_SHAPE, "vidc_enrich_scatter: bad shape");
    hipLaunchKernelGGL(enrich_scatter_kernel, dim3(B), dim3(NT), 0, vidc::as_stream(stream), plane_depth, sub, sub_offsets, HW, enriched);
    VIDC_CHECK_LAUNCH("enrich_scatter_kernel");
    return VIDC_OK;
}
