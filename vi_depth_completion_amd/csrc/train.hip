// Training step of the depth-completion network on the device: what `_run_training_iteration` (network_run.py:231-254)
// needs beyond the inference kernels -- BatchNorm in train() mode (batch statistics + running-statistic update) and its backward, the
// backward of ReLU / residual add / max-pool / bilinear upsampling / the padded 1x1 head, the weight gradients of the convolutions
// (wgrad: an fp32-MFMA GEMM whose reduction runs over the pixels), the masked L1 loss of `_network_loss` (network_run.py:158-173)
// and torch.optim.Adam's update (network_run.py:228-229).  The data gradients of the convolutions (dgrad) are the forward conv kernel
// (conv_mfma.hip) on flipped / transposed weights (pack_weight_dgrad_kernel here), with stride-2 layers going through a zero-stuffed
// copy of the incoming gradient.
//
// Layout: activations NHWC fp32, `ld` = channel stride of a row (a tensor may be a channel slice of a concat buffer).  Per-channel
// reductions accumulate in fp64 with a fixed two-level order (bit-reproducible), like torch's CPU kernels accumulate float in double.
#include "common.h"
#include <cstdint>

namespace {

constexpr int TT = 256;

// What the per-channel sums become, applied by the thread that finishes a channel in chan_final_kernel (no extra launch).
// batch mean, biased variance and invstd from the two per-channel sums (shared by chan_final_kernel's FinalStats and the kernels that
// fold that reduction into their prologue, so that both forms round identically)
struct BnMoments { double mu, var; float mean, rstd; };
__device__ inline BnMoments bn_moments(double s0, double s1, long long M, float eps) {
    BnMoments r;
    r.mu = s0 / (double)M;
    r.var = s1 / (double)M - r.mu * r.mu;
    if (r.var < 0.0) r.var = 0.0;
    r.mean = (float)r.mu;
    r.rstd = (float)(1.0 / sqrt(r.var + (double)eps));
    return r;
}
__device__ inline void bn_update_running(float* running_mean, float* running_var, int c, const BnMoments& m, long long M, float momentum) {
    const double unbiased = M > 1 ? m.var * (double)M / (double)(M - 1) : m.var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m.mu);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
}

struct FinalStats {      // nn.BatchNorm2d.forward in train(): batch mean, biased variance, invstd = 1/sqrt(var + eps); running_mean/var <-
    long long M;         // (1 - momentum) * old + momentum * (mean, unbiased var)
    float eps, momentum;
    float *mean, *rstd, *running_mean, *running_var;
    __device__ void operator()(int c, double s0, double s1) const {
        const BnMoments m = bn_moments(s0, s1, M, eps);
        mean[c] = m.mean;
        rstd[c] = m.rstd;
        if (running_mean) bn_update_running(running_mean, running_var, c, m, M, momentum);
    }
};
struct FinalParamGrad {  // dbeta = sum dy, dgamma = sum dy * xhat
    float *dgamma, *dbeta;
    __device__ void operator()(int c, double s0, double s1) const { dbeta[c] = (float)s0; dgamma[c] = (float)s1; }
};
struct FinalColsum {     // bias gradient of a conv: column sums of dY (first sum only)
    float* out;
    __device__ void operator()(int c, double s0, double) const { out[c] = (float)s0; }
};

// The chunk sums of a 64-channel column added up exactly as chan_final_kernel adds them -- lane j takes chunks j, j + 32, ... in order,
// then the fixed pairwise tree over the 32 lanes -- by the 256 threads of ONE workgroup: thread = channel l x 8 of the 32 lanes, the 16
// loads of a round issued together, the tree through LDS.  Result: red[q][0][l].  Used by the kernels that fold the final reduction into
// their prologue (bn_apply_fold_kernel, bn_bwd_apply_t64_kernel<true>).  Bit-identical to chan_final_kernel.
__device__ __forceinline__ void column_sums(const double* __restrict__ partial, int n_chunks, int C, int c0, double (*red)[32][64]) {
    const int l = threadIdx.x & 63, j0 = (threadIdx.x >> 6) * 8, cc = c0 + l;
    double t0[8], t1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t0[j] = 0.0; t1[j] = 0.0; }
    if (cc < C)
        for (int k0 = 0; k0 < n_chunks; k0 += 32) {
            double u0[8], u1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = min(k0 + j0 + j, n_chunks - 1);
                const double* src = partial + ((size_t)k * 2) * C + cc;
                u0[j] = src[0];
                u1[j] = src[C];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j0 + j < n_chunks) { t0[j] += u0[j]; t1[j] += u1[j]; }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][j0 + j][l] = t0[j]; red[1][j0 + j][l] = t1[j]; }
    __syncthreads();
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {            // chan_final_kernel's tree: lane j += lane j + w for j < w
        for (int i = threadIdx.x; i < w * 64; i += TT) {
            const int j = i >> 6, ll = i & 63;
            red[0][j][ll] += red[0][j + w][ll];
            red[1][j][ll] += red[1][j + w][ll];
        }
        __syncthreads();
    }
}

// ---- per-channel sums over rows: partial[chunk][q][C] (fp64) -------------------------------------------------------------------------
// block = 256 threads = 64 channels x 4 row-lanes; grid = (C/64 rounded up, n_chunks).  MODE 0: sum x, sum x^2.  MODE 1 (BN backward):
// sum dy', sum dy' * xhat with dy' = dy * (y > 0) when y != NULL (the ReLU that followed the BatchNorm).
template <int MODE>
__global__ void __launch_bounds__(TT)
chan_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ y, long long M, int C, int lda, int ldb,
                    int ldy, const float* __restrict__ mean, const float* __restrict__ rstd, int rows_per_chunk, double* __restrict__ partial) {
    // block = 16 channel quads (64 channels, one float4 per thread and row) x 16 row-lanes; C % 4 == 0 and the strides % 4 == 0 (host-checked)
    __shared__ double red[2][16][64];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (c < C) {
        float mu[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0};
        if (MODE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mu[k] = mean[c + k]; rs[k] = rstd[c + k]; }
        }
        for (long long r = r0 + rl; r < r1; r += 16) {
            const float4 av = *reinterpret_cast<const float4*>(a + r * lda + c);
            float g[4] = {av.x, av.y, av.z, av.w};
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double v = (double)g[k]; s0[k] += v; s1[k] += v * v; }
            } else {
                const float4 bv = *reinterpret_cast<const float4*>(b + r * ldb + c);
                const float xv[4] = {bv.x, bv.y, bv.z, bv.w};
                if (y) {
                    const float4 yv4 = *reinterpret_cast<const float4*>(y + r * ldy + c);
                    const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (!(yv[k] > 0.f)) g[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[k] - mu[k]) * rs[k];
                    s0[k] += (double)g[k]; s1[k] += (double)g[k] * (double)xh;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][rl][cq * 4 + k] = s0[k]; red[1][rl][cq * 4 + k] = s1[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {                       // 64 channels x 2 sums; row-lanes added in order 0..15
        const int q = threadIdx.x >> 6, l = threadIdx.x & 63, cc = blockIdx.x * 64 + l;
        if (cc < C) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) t += red[q][j][l];
            partial[((size_t)blockIdx.y * 2 + q) * C + cc] = t;
        }
    }
}

// sums[q][C] = sum over chunks: block = 8 channels x 32 lanes; lane l adds chunks l, l+32, ... in order, then the 32 lane sums are added
// pairwise (a fixed tree) -- bit-reproducible; `fin` then turns the two sums of a channel into the op's per-channel outputs
template <typename Final>
__global__ void __launch_bounds__(TT) chan_final_kernel(const double* __restrict__ partial, int n_chunks, int C, double* __restrict__ sums, Final fin) {
    __shared__ double red[2][32][8];
    const int l = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + l;
    double s0 = 0.0, s1 = 0.0;
    if (c < C)
        for (int k = rl; k < n_chunks; k += 32) { s0 += partial[((size_t)k * 2 + 0) * C + c]; s1 += partial[((size_t)k * 2 + 1) * C + c]; }
    red[0][rl][l] = s0; red[1][rl][l] = s1;
    __syncthreads();
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        if (rl < w) { red[0][rl][l] += red[0][rl + w][l]; red[1][rl][l] += red[1][rl + w][l]; }
        __syncthreads();
    }
    if (rl == 0 && c < C) {
        s0 = red[0][0][l];
        s1 = red[1][0][l];
        sums[c] = s0;
        sums[C + c] = s1;
        fin(c, s0, s1);
    }
}

// The kernels below fold the final reduction into their prologue -- every workgroup redoes it for its 64 channels from the L2-resident
// chunk sums (column_sums) -- which removes the chan_final launch between the partial-sum kernel and its consumer without any
// cross-workgroup hand-off.  Bit-identical.
// chan_final<FinalStats> + bn_apply_kernel in one launch: 64 pixels x 64 channels per workgroup (the layout of bn_bwd_apply_t64_kernel).
__global__ void __launch_bounds__(256)
bn_apply_fold_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int C, int ldx, int ldy, const double* __restrict__ partial, int n_chunks,
                     float eps, float momentum, float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ running_mean,
                     float* __restrict__ running_var, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                     unsigned short* __restrict__ y_bf16, const float* __restrict__ res, int ldr) {
    __shared__ double red[2][32][64];
    __shared__ float mean_s[64], rstd_s[64];
    const int c0 = blockIdx.y * 64, m0 = blockIdx.x * 64;
    column_sums(partial, n_chunks, C, c0, red);
    if (threadIdx.x < 64) {
        const int l = threadIdx.x, cc = c0 + l;
        const BnMoments mo = bn_moments(red[0][0][l], red[1][0][l], (long long)M, eps);
        mean_s[l] = mo.mean; rstd_s[l] = mo.rstd;
        if (blockIdx.x == 0 && cc < C) {         // one workgroup per channel block publishes what the backward and the module keep
            save_mean[cc] = mo.mean; save_rstd[cc] = mo.rstd;
            if (running_mean) bn_update_running(running_mean, running_var, cc, mo, (long long)M, momentum);
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = c0 + tx * 4;
    if (c >= C) return;
    float alpha[4], bb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        alpha[k] = rstd_s[tx * 4 + k] * gamma[c + k];
        bb[k] = beta[c + k] - mean_s[tx * 4 + k] * alpha[k];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = m0 + ty + 16 * rr;
        if (r >= M) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        float in[4] = {v.x, v.y, v.z, v.w}, out[4], add[4] = {0.f, 0.f, 0.f, 0.f};
        if (res) {
            const float4 q = *reinterpret_cast<const float4*>(res + (size_t)r * ldr + c);
            add[0] = q.x; add[1] = q.y; add[2] = q.z; add[3] = q.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o = in[k] * alpha[k] + bb[k];
            if (res) o = __fadd_rn(o, add[k]);
            out[k] = relu ? fmaxf(o, 0.f) : o;
        }
        *reinterpret_cast<float4*>(y + (size_t)r * ldy + c) = make_float4(out[0], out[1], out[2], out[3]);
        if (y_bf16)
            *reinterpret_cast<uint2*>(y_bf16 + (size_t)r * C + c) = make_uint2(vidc::bf16_rne(out[0]) | ((unsigned)vidc::bf16_rne(out[1]) << 16),
                                                                               vidc::bf16_rne(out[2]) | ((unsigned)vidc::bf16_rne(out[3]) << 16));
    }
}

// y = x * alpha + beta' with alpha = invstd * gamma, beta' = beta - mean * alpha (torch's batch_norm_cpu_transform_input), ReLU optional
__global__ void __launch_bounds__(TT)
bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long M, int C, int ldx, int ldy, const float* __restrict__ mean,
                const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                unsigned short* __restrict__ y_bf16, const float* __restrict__ res, int ldr) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    if (i >= M * c4) return;
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    float in[4] = {v.x, v.y, v.z, v.w}, out[4], add[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) {          // Bottleneck tail: relu(bn3(.) + identity) in the same pass (the BatchNorm value is rounded to fp32 before the sum, as when
        const float4 q = *reinterpret_cast<const float4*>(res + r * ldr + c);      // it was stored and read back by a separate add kernel)
        add[0] = q.x; add[1] = q.y; add[2] = q.z; add[3] = q.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float alpha = rstd[c + k] * gamma[c + k];
        const float bb = beta[c + k] - mean[c + k] * alpha;
        float o = in[k] * alpha + bb;
        if (res) o = __fadd_rn(o, add[k]);
        out[k] = relu ? fmaxf(o, 0.f) : o;
    }
    *reinterpret_cast<float4*>(y + r * ldy + c) = make_float4(out[0], out[1], out[2], out[3]);
    if (y_bf16)         // the bf16 operand copy the next conv of a VIDC_PREC_BF16 step reads (dense rows of C bf16), saving its cast launch
        *reinterpret_cast<uint2*>(y_bf16 + r * C + c) = make_uint2(vidc::bf16_rne(out[0]) | ((unsigned)vidc::bf16_rne(out[1]) << 16),
                                                                   vidc::bf16_rne(out[2]) | ((unsigned)vidc::bf16_rne(out[3]) << 16));
}

// dx = gamma * invstd * (dy' - sum(dy')/M - xhat * sum(dy' * xhat)/M); dgamma = sum(dy' * xhat); dbeta = sum(dy')
__global__ void __launch_bounds__(TT)
bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ dx, long long M, int C,
                    int lddy, int ldx, int ldy, int lddx, const float* __restrict__ mean, const float* __restrict__ rstd,
                    const float* __restrict__ gamma, const double* __restrict__ sums, unsigned short* __restrict__ dx_bf16) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    if (i >= M * c4) return;
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const float4 g4 = *reinterpret_cast<const float4*>(dy + r * lddy + c), x4 = *reinterpret_cast<const float4*>(x + r * ldx + c);
    float g[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w}, o[4];
    if (y) {
        const float4 y4 = *reinterpret_cast<const float4*>(y + r * ldy + c);
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (!(yv[k] > 0.f)) g[k] = 0.f;
    }
    const double invM = 1.0 / (double)M;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xh = (xv[k] - mean[c + k]) * rstd[c + k];
        const float m1 = (float)(sums[c + k] * invM), m2 = (float)(sums[C + c + k] * invM);
        o[k] = gamma[c + k] * rstd[c + k] * (g[k] - m1 - xh * m2);
    }
    *reinterpret_cast<float4*>(dx + r * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
    if (dx_bf16)
        *reinterpret_cast<uint2*>(dx_bf16 + r * C + c) = make_uint2(vidc::bf16_rne(o[0]) | ((unsigned)vidc::bf16_rne(o[1]) << 16),
                                                                    vidc::bf16_rne(o[2]) | ((unsigned)vidc::bf16_rne(o[3]) << 16));
}

// The same map on 64 pixels x 64 channels per workgroup, which ALSO writes the transpose of dx as plain bf16 rows [C][Mp] (zero for
// m >= M): dx of the BatchNorm behind a conv is that conv's dY, and its weight-gradient GEMM reads dY^T in exactly this format
// (vidc_im2col_transposed(..., KH = KW = 1, split = 2) -- one launch and one pass over dY per conv that this kernel makes unnecessary).
// LDS tile [pixel][channel], pitch 65 floats, both phases as in im2col_t64_kernel; every value is rounded once, from the fp32 result.
// FOLD: `sums` holds the CHUNK sums of chan_partial_kernel<1> (n_chunks of them) and the workgroup adds them up itself for its 64 channels
// (column_sums: chan_final's order); the workgroups of the first pixel block write dgamma / dbeta.  dx_bf16_t may then be NULL (no
// transposed copy wanted): the kernel is the folded form of bn_bwd_apply_kernel as well.
template <bool FOLD>
__global__ void __launch_bounds__(256)
bn_bwd_apply_t64_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ dx, int M, int C,
                        int lddy, int ldx, int ldy, int lddx, const float* __restrict__ mean, const float* __restrict__ rstd,
                        const float* __restrict__ gamma, const double* __restrict__ sums, unsigned short* __restrict__ dx_bf16,
                        unsigned short* __restrict__ dx_bf16_t, int Mp, int n_chunks, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // (FOLD: the reduction scratch [2][32][64] doubles and the transpose tile [64][65] floats share the same LDS, one after the other)
    __shared__ __attribute__((aligned(16))) unsigned char raw_s[FOLD ? 2 * 32 * 64 * 8 : 64 * 65 * 4];
    __shared__ double sum_s[2][64];
    float (*tile)[65] = reinterpret_cast<float (*)[65]>(raw_s);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int c = c0 + tx * 4;
    const double invM = 1.0 / (double)M;
    if (FOLD) {
        double (*red)[32][64] = reinterpret_cast<double (*)[32][64]>(raw_s);
        column_sums(sums, n_chunks, C, c0, red);
        if (threadIdx.x < 128) {
            const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
            const double t = c0 + l < C ? red[q][0][l] : 0.0;
            sum_s[q][l] = t;
            if (blockIdx.x == 0 && c0 + l < C) (q ? dgamma : dbeta)[c0 + l] = (float)t;      // FinalParamGrad: dbeta = sum dy', dgamma = sum dy' * xhat
        }
        __syncthreads();
    }
    float mu[4], rs[4], gs[4], m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ck = c + k < C ? c + k : C - 1;
        mu[k] = mean[ck]; rs[k] = rstd[ck]; gs[k] = gamma[ck] * rs[k];
        const double t0 = FOLD ? sum_s[0][ck - c0] : sums[ck], t1 = FOLD ? sum_s[1][ck - c0] : sums[C + ck];
        m1[k] = (float)(t0 * invM); m2[k] = (float)(t1 * invM);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + ty + 16 * r;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < M && c < C) {
            const float4 g4 = *reinterpret_cast<const float4*>(dy + (size_t)m * lddy + c), x4 = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + c);
            float g[4] = {g4.x, g4.y, g4.z, g4.w};
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
            if (y) {
                const float4 y4 = *reinterpret_cast<const float4*>(y + (size_t)m * ldy + c);
                const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) if (!(yv[k] > 0.f)) g[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xv[k] - mu[k]) * rs[k];
                o[k] = gs[k] * (g[k] - m1[k] - xh * m2[k]);
            }
            if (dx) *reinterpret_cast<float4*>(dx + (size_t)m * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);      // (NULL: nobody reads the fp32 form)
            if (dx_bf16)
                *reinterpret_cast<uint2*>(dx_bf16 + (size_t)m * C + c) = make_uint2(vidc::bf16_rne(o[0]) | ((unsigned)vidc::bf16_rne(o[1]) << 16),
                                                                                    vidc::bf16_rne(o[2]) | ((unsigned)vidc::bf16_rne(o[3]) << 16));
        }
        float* t = &tile[ty + 16 * r][tx * 4];
        t[0] = o[0]; t[1] = o[1]; t[2] = o[2]; t[3] = o[3];
    }
    if (!dx_bf16_t) return;               // (uniform over the launch)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cl = ty + 16 * r, cc = c0 + cl, m = m0 + tx * 4;
        if (cc < C && m < Mp) {
            const float v[4] = {tile[tx * 4][cl], tile[tx * 4 + 1][cl], tile[tx * 4 + 2][cl], tile[tx * 4 + 3][cl]};
            *reinterpret_cast<uint2*>(dx_bf16_t + (size_t)cc * Mp + m) = make_uint2((unsigned)vidc::bf16_rne(v[0]) | ((unsigned)vidc::bf16_rne(v[1]) << 16),
                                                                                    (unsigned)vidc::bf16_rne(v[2]) | ((unsigned)vidc::bf16_rne(v[3]) << 16));
        }
    }
}

// ---- elementwise ---------------------------------------------------------------------------------------------------------------
// y = relu?(a + b)  (Bottleneck: out = relu(bn3(conv3) + identity); decoder: z1 + z2 + z3 + z4 without ReLU)
__global__ void __launch_bounds__(TT)
add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long M, int C, int lda, int ldb, int ldy, int relu,
           unsigned short* __restrict__ y_bf16) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    if (i >= M * c4) return;
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const float4 u = *reinterpret_cast<const float4*>(a + r * lda + c), v = *reinterpret_cast<const float4*>(b + r * ldb + c);
    float4 o = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + r * ldy + c) = o;
    if (y_bf16)         // the bf16 operand copy (dense rows) the convs of the next block read in a VIDC_PREC_BF16 step: saves their cast launches
        *reinterpret_cast<uint2*>(y_bf16 + r * C + c) = make_uint2(vidc::bf16_rne(o.x) | ((unsigned)vidc::bf16_rne(o.y) << 16),
                                                                   vidc::bf16_rne(o.z) | ((unsigned)vidc::bf16_rne(o.w) << 16));
}

// dx (=|+=) dy * (y > 0)   (y NULL: plain copy / accumulate)
__global__ void __launch_bounds__(TT)
relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long M, int C, int lddy, int ldy, int lddx, int accumulate) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    if (i >= M * c4) return;
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + c);
    if (y) {
        const float4 v = *reinterpret_cast<const float4*>(y + r * ldy + c);
        if (!(v.x > 0.f)) g.x = 0.f;
        if (!(v.y > 0.f)) g.y = 0.f;
        if (!(v.z > 0.f)) g.z = 0.f;
        if (!(v.w > 0.f)) g.w = 0.f;
    }
    float4* d = reinterpret_cast<float4*>(dx + r * lddx + c);
    if (accumulate) { const float4 o = *d; g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
    *d = g;
}

// ---- max-pool 3x3 / stride 2 / pad 1 backward (gather, deterministic) -------------------------------------------------------------
// An input pixel receives dy of every window whose FIRST maximum (scan order kh, kw like torch's CPU kernel) it is.
__global__ void __launch_bounds__(TT)
maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C, int ldx, int lddy, int lddx) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    const long long total = (long long)B * H * W * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    float acc = 0.f;
    for (int oy = h / 2; oy <= (h + 1) / 2; ++oy) {       // the windows with 2*oy-1 <= h <= 2*oy+1: one for even h, two for odd h
        if (oy >= Ho) continue;
        for (int ox = w / 2; ox <= (w + 1) / 2; ++ox) {
            if (ox >= Wo) continue;
            float best = -INFINITY;
            int bh = -1, bw = -1;
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = 2 * oy - 1 + kh;
                if (ih < 0 || ih >= H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = 2 * ox - 1 + kw;
                    if (iw < 0 || iw >= W) continue;
                    const float v = x[(((long long)b * H + ih) * W + iw) * ldx + c];
                    if (v > best || bh < 0) { best = v; bh = ih; bw = iw; }
                }
            }
            if (bh == h && bw == w) acc += dy[(((long long)b * Ho + oy) * Wo + ox) * lddy + c];
        }
    }
    dx[(((long long)b * H + h) * W + w) * lddx + c] = acc;
}
// The same map, four channels per thread with 16-byte accesses and 32-bit offsets (C, ldx, lddy, lddx multiples of 4, tensors below 2^31
// elements: the pyramids' 64-channel pool); per channel the scan order and the additions are those of the scalar kernel: identical bits.
__global__ void __launch_bounds__(TT)
maxpool_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C, int ldx, int lddy, int lddx) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, c4 = C >> 2;
    const unsigned i = blockIdx.x * TT + threadIdx.x;
    if (i >= (unsigned)B * H * W * c4) return;
    const int c = (int)(i % c4) * 4;
    unsigned p = i / c4;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oy = h / 2; oy <= (h + 1) / 2; ++oy) {
        if (oy >= Ho) continue;
        for (int ox = w / 2; ox <= (w + 1) / 2; ++ox) {
            if (ox >= Wo) continue;
            float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int at[4] = {-1, -1, -1, -1};                    // kh * 3 + kw of the first maximum
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = 2 * oy - 1 + kh;
                if (ih < 0 || ih >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = 2 * ox - 1 + kw;
                    if (iw < 0 || iw >= W) continue;
                    const float4 v4 = *reinterpret_cast<const float4*>(x + (unsigned)((b * H + ih) * W + iw) * (unsigned)ldx + c);
                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (v[k] > best[k] || at[k] < 0) { best[k] = v[k]; at[k] = kh * 3 + kw; }
                }
            }
            const int mine = (h - (2 * oy - 1)) * 3 + (w - (2 * ox - 1));
            const float4 g4 = *reinterpret_cast<const float4*>(dy + (unsigned)((b * Ho + oy) * Wo + ox) * (unsigned)lddy + c);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) if (at[k] == mine) acc[k] += g[k];
        }
    }
    *reinterpret_cast<float4*>(dx + (unsigned)((b * H + h) * W + w) * (unsigned)lddx + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ---- bilinear upsampling (align_corners = True) backward, gather form ------------------------------------------------------------
// Forward (pointwise.hip / F.interpolate): src = dst * (in - 1) / (out - 1) in fp32, i0 = floor(src), i1 = min(i0 + 1, in - 1),
// lambda = src - i0.  The input pixel (i, j) collects every output pixel that interpolated from it.
__device__ inline void up_src(int d, int in, int out, int& i0, int& i1, float& lam) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float s = scale * (float)d;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    lam = s - (float)i0;
}
__global__ void __launch_bounds__(TT)
upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int h, int w, int C, int lddy, int lddx, int H, int W) {
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    const long long total = (long long)B * h * w * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int jj = (int)(p % w); p /= w;
    const int ii = (int)(p % h);
    const int b = (int)(p / h);
    // candidate output rows: src in (ii - 1, ii + 1)  ->  dst in ((ii-1)/scale, (ii+1)/scale); scanned with a margin of one
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    int Y0 = 0, Y1 = H - 1, X0 = 0, X1 = W - 1;
    if (sy > 0.f) { Y0 = max(0, (int)floorf((float)(ii - 1) / sy) - 1); Y1 = min(H - 1, (int)ceilf((float)(ii + 1) / sy) + 1); }
    if (sx > 0.f) { X0 = max(0, (int)floorf((float)(jj - 1) / sx) - 1); X1 = min(W - 1, (int)ceilf((float)(jj + 1) / sx) + 1); }
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        int a0, a1; float ly;
        up_src(Y, h, H, a0, a1, ly);
        const float wy = (a0 == ii ? 1.f - ly : 0.f) + (a1 == ii ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            int b0, b1; float lx;
            up_src(X, w, W, b0, b1, lx);
            const float wx = (b0 == jj ? 1.f - lx : 0.f) + (b1 == jj ? lx : 0.f);
            if (wx == 0.f) continue;
            acc += wy * wx * dy[(((long long)b * H + Y) * W + X) * lddy + c];
        }
    }
    dx[(((long long)b * h + ii) * w + jj) * lddx + c] = acc;
}
// The same map, four channels per thread (16-byte accesses, 32-bit offsets): the interpolation weights of a pixel are computed once for
// the four; per channel the (Y, X) scan order and every product are those of the scalar kernel -- identical bits.
__global__ void __launch_bounds__(TT)
upsample_bwd4_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int h, int w, int C, int lddy, int lddx, int H, int W) {
    const int c4 = C >> 2;
    const unsigned i = blockIdx.x * TT + threadIdx.x;
    if (i >= (unsigned)B * h * w * c4) return;
    const int c = (int)(i % c4) * 4;
    unsigned p = i / c4;
    const int jj = (int)(p % w); p /= w;
    const int ii = (int)(p % h);
    const int b = (int)(p / h);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    int Y0 = 0, Y1 = H - 1, X0 = 0, X1 = W - 1;
    if (sy > 0.f) { Y0 = max(0, (int)floorf((float)(ii - 1) / sy) - 1); Y1 = min(H - 1, (int)ceilf((float)(ii + 1) / sy) + 1); }
    if (sx > 0.f) { X0 = max(0, (int)floorf((float)(jj - 1) / sx) - 1); X1 = min(W - 1, (int)ceilf((float)(jj + 1) / sx) + 1); }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int Y = Y0; Y <= Y1; ++Y) {
        int a0, a1; float ly;
        up_src(Y, h, H, a0, a1, ly);
        const float wy = (a0 == ii ? 1.f - ly : 0.f) + (a1 == ii ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            int b0, b1; float lx;
            up_src(X, w, W, b0, b1, lx);
            const float wx = (b0 == jj ? 1.f - lx : 0.f) + (b1 == jj ? lx : 0.f);
            if (wx == 0.f) continue;
            const float4 g = *reinterpret_cast<const float4*>(dy + (unsigned)((b * H + Y) * W + X) * (unsigned)lddy + c);
            const float ww = wy * wx;
            acc[0] += ww * g.x; acc[1] += ww * g.y; acc[2] += ww * g.z; acc[3] += ww * g.w;
        }
    }
    *reinterpret_cast<float4*>(dx + (unsigned)((b * h + ii) * w + jj) * (unsigned)lddx + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ---- padded 1x1 head (depth_completion.py:141-147: Conv2d(192, 1, 1, padding=1)) backward -----------------------------------------
// g_low: [B][h+2][w+2] gradient at the 62x82 map; x: NHWC [B][h][w][C].  dx[b,y,x,c] = g_low[b,y+1,x+1] * w[c]
__global__ void __launch_bounds__(TT)
head_dgrad_kernel(const float* __restrict__ g_low, const float* __restrict__ wgt, float* __restrict__ dx, int B, int h, int w, int C, int lddx) {
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    const long long total = (long long)B * h * w * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int xx = (int)(p % w); p /= w;
    const int yy = (int)(p % h);
    const int b = (int)(p / h);
    dx[(((long long)b * h + yy) * w + xx) * lddx + c] = g_low[((long long)b * (h + 2) + yy + 1) * (w + 2) + xx + 1] * wgt[c];
}
// dw[c] = sum g_low_interior * x[...,c]; one block per channel chunk would re-read g: here one thread per (chunk, c), partials in fp64
__global__ void __launch_bounds__(TT)
head_wgrad_partial_kernel(const float* __restrict__ g_low, const float* __restrict__ x, int B, int h, int w, int C, int ldx, int rows_per_chunk,
                          double* __restrict__ partial) {
    const int c = blockIdx.x * TT + threadIdx.x;
    if (c >= C) return;
    const long long M = (long long)B * h * w;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    double s = 0.0;
    for (long long r = r0; r < r1; ++r) {
        const int xx = (int)(r % w);
        const long long q = r / w;
        const int yy = (int)(q % h), b = (int)(q / h);
        s += (double)g_low[((long long)b * (h + 2) + yy + 1) * (w + 2) + xx + 1] * (double)x[r * ldx + c];
    }
    partial[(size_t)blockIdx.y * C + c] = s;
}
// kFinalLanes lanes per output: lane j sums chunks j, j + kFinalLanes, ... and the partial sums are added in the order j = 0, 1, ... -- a
// fixed order, so the result is reproducible; one lane per output walked up to 600 chunks serially (120 us on the critical path of every
// pyramid's stem).
constexpr int kFinalLanes = 32;
__global__ void __launch_bounds__(TT) head_wgrad_final_kernel(const double* __restrict__ partial, int n_chunks, int C, float* __restrict__ dw) {
    __shared__ double red[kFinalLanes][TT / kFinalLanes];
    const int l = threadIdx.x % (TT / kFinalLanes), j = threadIdx.x / (TT / kFinalLanes);
    const int c = blockIdx.x * (TT / kFinalLanes) + l;
    double s = 0.0;
    if (c < C)
        for (int k = j; k < n_chunks; k += kFinalLanes) s += partial[(size_t)k * C + c];
    red[j][l] = s;
    __syncthreads();
    if (j == 0 && c < C) {
        double t = red[0][l];
#pragma unroll
        for (int q = 1; q < kFinalLanes; ++q) t += red[q][l];
        dw[c] = (float)t;
    }
}

// ---- generic fixed-order sum of n floats (fp64): head bias gradient, loss ----------------------------------------------------------
__global__ void __launch_bounds__(TT) sum_partial_kernel(const float* __restrict__ v, long long n, int per, double* __restrict__ partial) {
    __shared__ double red[TT];
    const long long base = (long long)blockIdx.x * TT * per;
    double s = 0.0;
    for (int k = 0; k < per; ++k) {
        const long long i = base + (long long)k * TT + threadIdx.x;
        if (i < n) s += (double)v[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = TT / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(TT) sum_final_kernel(const double* __restrict__ partial, int n, double* __restrict__ out_d, float* __restrict__ out_f) {
    __shared__ double red[TT];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += TT) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = TT / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (out_d) *out_d = red[0]; if (out_f) *out_f = (float)red[0]; }
}

// ---- masked L1 loss (network_run.py:163-173): terms |pred - gt| / (H*W) on gt > 0, and d loss / d pred ----------------------------
__global__ void __launch_bounds__(TT)
l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long long n, float inv_hw, float* __restrict__ terms, float* __restrict__ dpred) {
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    if (i >= n) return;
    const float g = gt[i], d = pred[i] - g;
    const bool on = g > 0.f;
    terms[i] = on ? fabsf(d) * inv_hw : 0.f;
    dpred[i] = on ? (d > 0.f ? inv_hw : (d < 0.f ? -inv_hw : 0.f)) : 0.f;
}

// ---- Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) on a flat parameter buffer -------------------------------------
// One element; every product / sum / quotient rounds once (contraction off: HIP's __fmul_rn / __fadd_rn are plain operators that hipcc
// may still fuse), so an element's bits do not depend on whether the four-wide body or the scalar tail of the launch handled it (the flat
// layout -- and with it an element's position -- differs between the grouped and the per-pyramid form of the step, and
// tests/test_training.py compares their stepped parameters bit for bit).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float c1, float c2, float eps, float step_size,
                                         float bc2_sqrt) {
#pragma clang fp contract(off)
    m = m * b1 + g * c1;
    v = v * b2 + (g * g) * c2;
    const float denom = __fsqrt_rn(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}
// 28 bytes per parameter (p, g, m, v in; p, m, v out): one thread per four consecutive parameters (16-byte accesses), the n % 4 tail scalar
__global__ void __launch_bounds__(TT)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n, float lr, float b1, float b2,
            float eps, float bc1, float bc2_sqrt) {
    const long long i = ((long long)blockIdx.x * TT + threadIdx.x) * 4;
    if (i >= n) return;
    const float c1 = 1.f - b1, c2 = 1.f - b2, step_size = lr / bc1;
    if (i + 4 <= n) {
        float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        adam_one(pv.x, gv.x, mv.x, vv.x, b1, b2, c1, c2, eps, step_size, bc2_sqrt);
        adam_one(pv.y, gv.y, mv.y, vv.y, b1, b2, c1, c2, eps, step_size, bc2_sqrt);
        adam_one(pv.z, gv.z, mv.z, vv.z, b1, b2, c1, c2, eps, step_size, bc2_sqrt);
        adam_one(pv.w, gv.w, mv.w, vv.w, b1, b2, c1, c2, eps, step_size, bc2_sqrt);
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
        *reinterpret_cast<float4*>(p + i) = pv;
    } else {
        for (long long k = i; k < n; ++k) adam_one(p[k], g[k], m[k], v[k], b1, b2, c1, c2, eps, step_size, bc2_sqrt);
    }
}

// ---- dgrad helpers -------------------------------------------------------------------------------------------------------------
// Weights of the dgrad convolution, packed for conv_mfma.hip: the conv computing dX from dY has Cout' = Cin, Cin' = Cout and the
// kernel flipped: wp[ci][co/32][KH-1-kh][KW-1-kw][co%32] = w[co][ci][kh][kw]  (wp is [Cout'][Cin'/32][KH][KW][32]).
__global__ void __launch_bounds__(TT)
pack_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KH, int KW) {
    const long long total = (long long)Cout * Cin * KH * KW;
    const long long idx = (long long)blockIdx.x * TT + threadIdx.x;
    if (idx >= total) return;
    const int Kp = Cout * KH * KW;                     // K of the dgrad conv
    const int ci = (int)(idx / Kp), k = (int)(idx - (long long)ci * Kp);
    const int lane = k & 31, u = k >> 5, taps = KH * KW;
    const int cu = u / taps, tap = u - cu * taps, kh = tap / KW, kw = tap - kh * KW, co = cu * 32 + lane;
    wp[idx] = w[(((long long)co * Cin + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
}
// Every conv weight of a network re-packed in ONE launch (the parameters move every optimizer step): block -> item by binary search over
// items[].block_begin (uniform per block: scalar loads).  Both maps are staged through LDS so that the parameter is read and the packed
// copy written in contiguous runs:
//  forward (kind & 1 == 0): a U-channel K unit of a row is the same U*taps contiguous elements in the parameter ([U][taps]) and in the
//      packed row ([taps][U]) -- a local transpose; a block does kPackUnits of them (for a 1x1 conv it is a copy);
//  dgrad (kind & 1): a block takes U output channels x kPackCi input channels: U contiguous runs of kPackCi*taps parameter elements in,
//      kPackCi*taps runs of U packed elements out (kernel flipped).
// U = 32 channels per 128-byte unit for fp32 / split-bf16 ([32 x hi | 32 x lo]), 64 for plain bf16.
// (a 1x1 conv has 1/9 of the elements per unit / per input channel: its blocks take 64 of them instead of 8 -- with 8 a block moved 256-512
//  elements and the launch, ~600 000 blocks for the 310 M parameters, was bound by block start-up and the item search, not by bandwidth)
constexpr int kPackUnits = 8, kPackCi = 8;
__host__ __device__ inline int pack_units(int taps) { return taps == 1 ? 64 : kPackUnits; }
__host__ __device__ inline int pack_ci(int taps) { return taps == 1 ? 64 : kPackCi; }
__host__ __device__ inline long long pack_item_blocks(int Cout, int Cin, int KH, int KW, int kind) {
    const int U = (kind & 4) ? 64 : 32, pu = pack_units(KH * KW), pc = pack_ci(KH * KW);
    if (!(kind & 1)) return ((long long)Cout * (Cin / U) + pu - 1) / pu;
    return (long long)(Cout / U) * ((Cin + pc - 1) / pc);
}
__device__ inline void pack_store(const vidc_pack_item& it, long long off, float v) {
    if (it.kind & 4) {
        reinterpret_cast<unsigned short*>(it.packed)[off] = vidc::bf16_rne(v);
    } else if (it.kind & 2) {
        unsigned short h, l;
        vidc::split_bf16(v, h, l);
        unsigned short* base = reinterpret_cast<unsigned short*>(it.packed) + (off >> 5) * 64 + (off & 31);
        base[0] = h;
        base[32] = l;
    } else {
        it.packed[off] = v;
    }
}
// Eight consecutive packed elements rounded to bf16 and written by one 16-byte store (dst 16-byte aligned).
__device__ inline void store8_bf16(unsigned short* __restrict__ dst, const float* v) {
    uint4 o;
    o.x = (unsigned)vidc::bf16_rne(v[0]) | ((unsigned)vidc::bf16_rne(v[1]) << 16);
    o.y = (unsigned)vidc::bf16_rne(v[2]) | ((unsigned)vidc::bf16_rne(v[3]) << 16);
    o.z = (unsigned)vidc::bf16_rne(v[4]) | ((unsigned)vidc::bf16_rne(v[5]) << 16);
    o.w = (unsigned)vidc::bf16_rne(v[6]) | ((unsigned)vidc::bf16_rne(v[7]) << 16);
    *reinterpret_cast<uint4*>(dst) = o;
}
// Round 4, plain-bf16 kinds (the training mode configs[4] names; 1.41 ms for the 310 M parameters before): a thread produces eight
// consecutive bf16 values and stores them at once (2-byte stores before), and a 1x1 forward copy -- the same order as the parameter --
// skips the LDS round trip: 0.79 ms (4.7 TB/s).  Same values.  (A block -> item index instead of the search over block_begin was
// measured too: 0.79 against 0.76 ms, nothing -- the table is cache-resident and the search overlaps other blocks' traffic.)
__global__ void __launch_bounds__(TT) pack_batched_kernel(const vidc_pack_item* __restrict__ items, int n) {
    __shared__ float tile[64 * (kPackCi * 9 + 1)];          // 64 channels x (8 x 9 + 1) floats = 18.25 KB; also holds 8 units x 64 x 9, 64 units x 64 x 1
    static_assert(64 * 64 <= 64 * (kPackCi * 9 + 1) && 64 * (64 + 1) <= 64 * (kPackCi * 9 + 1), "1x1 blocks must fit the tile");
    const long long blk = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].block_begin <= blk) lo = mid; else hi = mid - 1;
    }
    const vidc_pack_item it = items[lo];
    const int lb = (int)(blk - it.block_begin);
    const int taps = it.KH * it.KW, U = (it.kind & 4) ? 64 : 32, seg = U * taps;
    const bool plain = (it.kind & 4) != 0;                  // plain bf16: U == 64
    if (!(it.kind & 1)) {
        const int pu = pack_units(taps);
        const long long units = (long long)it.Cout * (it.Cin / U), u0 = (long long)lb * pu;
        const int cnt = (int)min((long long)pu, units - u0) * seg;
        const float* src = it.w + u0 * seg;
        const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;       // (a parameter sits anywhere in the flat buffer)
        if (plain && taps == 1) {                           // cnt % 64 == 0
            unsigned short* dst = reinterpret_cast<unsigned short*>(it.packed) + u0 * seg;
            for (int i = threadIdx.x * 8; i < cnt; i += TT * 8) {
                float v[8];
                if (aligned) {
                    const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = src[i + k];
                }
                store8_bf16(dst + i, v);
            }
            return;
        }
        if (aligned) {                                      // cnt % 4 == 0 (seg is a multiple of 32)
            for (int i = threadIdx.x * 4; i < cnt; i += TT * 4) *reinterpret_cast<float4*>(tile + i) = *reinterpret_cast<const float4*>(src + i);
        } else {
            for (int i = threadIdx.x; i < cnt; i += TT) tile[i] = src[i];
        }
        __syncthreads();
        if (plain) {                                        // eight lanes of one tap per thread
            unsigned short* dst = reinterpret_cast<unsigned short*>(it.packed) + u0 * seg;
            for (int i = threadIdx.x * 8; i < cnt; i += TT * 8) {
                const int u = i / seg, r = i - u * seg, tap = r >> 6, lane = r & 63;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[u * seg + (lane + k) * taps + tap];
                store8_bf16(dst + i, v);
            }
        } else {
            for (int i = threadIdx.x; i < cnt; i += TT) {
                const int u = i / seg, r = i - u * seg, tap = r / U, lane = r - tap * U;
                pack_store(it, u0 * seg + i, tile[u * seg + lane * taps + tap]);
            }
        }
    } else {
        const int pc = pack_ci(taps);
        const int ci_tiles = (it.Cin + pc - 1) / pc;
        const int cu = lb / ci_tiles, ci0 = (lb - cu * ci_tiles) * pc, nci = min(pc, it.Cin - ci0);
        const int run = nci * taps, pitch = pc * taps + 1;
        if (run == 64) {                                    // (the usual 1x1 block: no division)
            for (int i = threadIdx.x; i < U * 64; i += TT) {
                const int col = i >> 6, r = i & 63;
                tile[col * pitch + r] = it.w[((long long)(cu * U + col) * it.Cin + ci0) * taps + r];
            }
        } else {
            for (int i = threadIdx.x; i < U * run; i += TT) {
                const int col = i / run, r = i - col * run;
                tile[col * pitch + r] = it.w[((long long)(cu * U + col) * it.Cin + ci0) * taps + r];
            }
        }
        __syncthreads();
        const long long Kp = (long long)it.Cout * taps;
        if (plain) {                                        // eight output channels of one (input channel, tap) per thread
            unsigned short* dst = reinterpret_cast<unsigned short*>(it.packed);
            for (int i = threadIdx.x * 8; i < run * U; i += TT * 8) {
                const int cl = i / seg, r = i - cl * seg, tap = r >> 6, lane = r & 63;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[(lane + k) * pitch + cl * taps + (taps - 1 - tap)];
                store8_bf16(dst + (long long)(ci0 + cl) * Kp + (long long)(cu * taps + tap) * U + lane, v);
            }
        } else {
            for (int i = threadIdx.x; i < run * U; i += TT) {
                const int cl = i / seg, r = i - cl * seg, tap = r / U, lane = r - tap * U;
                pack_store(it, (long long)(ci0 + cl) * Kp + (long long)(cu * taps + tap) * U + lane, tile[lane * pitch + cl * taps + (taps - 1 - tap)]);
            }
        }
    }
}

// fp32 NHWC rows [rows][ldx] (C channels used) -> dense bf16 rows [rows][C] (round to nearest even); one thread per 8 channels
__global__ void __launch_bounds__(TT) cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long rows, int C, int ldx) {
    const int c8 = C / 8;
    const long long idx = (long long)blockIdx.x * TT + threadIdx.x;
    if (idx >= rows * c8) return;
    const long long r = idx / c8;
    const int c = (int)(idx - r * c8) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c), v1 = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
    const unsigned a = vidc::bf16_rne(v0.x) | ((unsigned)vidc::bf16_rne(v0.y) << 16), b = vidc::bf16_rne(v0.z) | ((unsigned)vidc::bf16_rne(v0.w) << 16);
    const unsigned d = vidc::bf16_rne(v1.x) | ((unsigned)vidc::bf16_rne(v1.y) << 16), e = vidc::bf16_rne(v1.z) | ((unsigned)vidc::bf16_rne(v1.w) << 16);
    *reinterpret_cast<uint4*>(y + r * C + c) = make_uint4(a, b, d, e);
}

// z[b, oy*s, ox*s, :] = dy[b, oy, ox, :], zero elsewhere (z is B x H x W x C, dense)
__global__ void __launch_bounds__(TT)
zero_stuff_kernel(const float* __restrict__ dy, float* __restrict__ z, int B, int Ho, int Wo, int C, int lddy, int s, int H, int W) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    const long long total = (long long)B * H * W * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    long long p = i / c4;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y % s == 0 && x % s == 0 && y / s < Ho && x / s < Wo)
        v = *reinterpret_cast<const float4*>(dy + (((long long)b * Ho + y / s) * Wo + x / s) * lddy + c);
    *reinterpret_cast<float4*>(z + (((long long)b * H + y) * W + x) * C + c) = v;
}

// ---- wgrad: dW[co][ci][kh][kw] = sum_m dY[m][co] * X[pixel(m) shifted by the tap][ci] on the fp32 matrix cores ---------------------
// v_mfma_f32_32x32x2_f32 with the reduction index = pixel: lane l supplies A[i = l & 31][k = l >> 5] = dY[m0 + (l >> 5)][co0 + (l & 31)]
// and B[k][j] = X[.. m0 + (l >> 5) ..][ci0 + (l & 31)]: both are 32 consecutive channels of an NHWC row, i.e. coalesced 128-byte
// loads straight from global memory, no transposition, no LDS.  One wave = one 32x32 (co, ci) tile of one tap over a range of pixels;
// a workgroup = 4 waves = 2x2 tiles (64 co x 64 ci); grid = (co tiles, ci tiles * taps, pixel chunks).  Partials of the pixel chunks
// are summed in chunk order by wgrad_final_kernel (deterministic), which also writes the OIHW layout of the parameter.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// One wave = a 64 x 64 (co, ci) register tile (2 x 2 MFMA tiles: 4 loads feed 4 MFMAs); a workgroup = 4 waves = 128 co x 128 ci.
__global__ void __launch_bounds__(256)
wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int lddy,
             int KH, int KW, int stride, int pad, int rows_per_chunk, float* __restrict__ partial /* [chunk][tap][Cout][Cin] */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int taps = KH * KW;
    const int ci_tiles = (Cin + 127) / 128;
    const int tap = blockIdx.y / ci_tiles, cit = blockIdx.y - tap * ci_tiles;
    const int kh = tap / KW, kw = tap - kh * KW;
    const int co0 = blockIdx.x * 128 + (wave >> 1) * 64, ci0 = cit * 128 + (wave & 1) * 64;
    const int co[2] = {co0 + li, co0 + 32 + li}, ci[2] = {ci0 + li, ci0 + 32 + li};
    const long long M = (long long)B * Ho * Wo;
    const long long m_begin = (long long)blockIdx.z * rows_per_chunk, m_end = min(M, m_begin + rows_per_chunk);
    f32x16 acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;
    // The pixel index of a lane advances by fixed steps, so (b, oy, ox) of the chunk's first pixel is decoded ONCE (wave-uniform: scalar
    // unit) and every lane walks from there with compares -- the first version divided twice per lane and pixel in 64-bit arithmetic,
    // ~10x the instructions of the MFMAs they fed.  Offsets are 32-bit (the host checks that both tensors stay below 2^31 elements).
    const int m_lo = (int)m_begin, m_hi = (int)m_end;
    int ox0 = m_lo % Wo, oy0 = (m_lo / Wo) % Ho, b0 = m_lo / (Wo * Ho);
    for (int m0 = m_lo; m0 < m_hi; m0 += 8) {           // 4 pixel pairs x (2 x 2) MFMAs per iteration, 16 loads in flight per lane
        float a[4][2], bv[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = m0 + 2 * t + lh;
            int ox = ox0 + 2 * t + lh, oy = oy0, b = b0;
            while (ox >= Wo) { ox -= Wo; ++oy; }
            while (oy >= Ho) { oy -= Ho; ++b; }
            a[t][0] = a[t][1] = bv[t][0] = bv[t][1] = 0.f;
            if (m < m_hi) {
                const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
                const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int xo = ((b * H + iy) * W + ix) * ldx;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (co[p] < Cout) a[t][p] = dy[m * lddy + co[p]];
                    if (inside && ci[p] < Cin) bv[t][p] = x[xo + ci[p]];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][p], bv[t][q], acc[p][q], 0, 0, 0);
        ox0 += 8;
        while (ox0 >= Wo) { ox0 -= Wo; ++oy0; }
        while (oy0 >= Ho) { oy0 -= Ho; ++b0; }
    }
    // C/D layout: col j = lane & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* out = partial + ((size_t)blockIdx.z * taps + tap) * (size_t)Cout * Cin;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cj = ci0 + q * 32 + li;
            if (cj >= Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = co0 + p * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ro < Cout) out[(size_t)ro * Cin + cj] = acc[p][q][r];
            }
        }
}
__global__ void __launch_bounds__(TT)
wgrad_final_kernel(const float* __restrict__ partial, int n_chunks, int taps, int Cout, int Cin, float* __restrict__ dw /* OIHW */) {
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;            // index into [tap][Cout][Cin]
    const long long per = (long long)taps * Cout * Cin;
    if (i >= per) return;
    double s = 0.0;
    for (int k = 0; k < n_chunks; ++k) s += (double)partial[(size_t)k * per + i];
    const int ci = (int)(i % Cin);
    const long long q = i / Cin;
    const int co = (int)(q % Cout), tap = (int)(q / Cout);
    dw[((long long)co * Cin + ci) * taps + tap] = (float)s;
}

// ---- wgrad as a GEMM on the conv kernel: transposed operands ---------------------------------------------------------------------------
// dW[co][tap][ci] = sum_m dY^T[co][m] * Xt[tap*C + ci][m] is a plain "both operands K-contiguous" GEMM with K = the pixels -- exactly
// the 1x1 case of conv_mfma.hip (activations = rows of dY^T, weights = rows of Xt), which runs at several times the rate of the
// direct kernel above (LDS-tiled, split-K, fp32 or bf16x3).  This kernel builds the operands: xt[(tap*C + c)][m] = x[b, oy*s - p + kh,
// ox*s - p + kw, c] (0 outside the image and for m >= M; rows are Mp = M rounded up to 32 long), through a 32x32 LDS transpose so that
// both the NHWC reads and the row writes are 128-byte coalesced.  With KH = KW = 1, s = 1, p = 0 it is the transpose of dY.  split != 0 writes
// the rows in the split-bf16 operand format of the bf16x3 mode directly (vidc_split_bf16x3's layout; same bytes per row).
__global__ void __launch_bounds__(256)
im2col_t_kernel(const float* __restrict__ x, float* __restrict__ xt, int B, int H, int W, int C, int ldx, int Ho, int Wo, int KH, int KW, int stride,
                int pad, int M, int Mp, int split) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z;
    const int kh = tap / KW, kw = tap - kh * KW;
    const bool chan_major = split & 4;      // rows ordered c * taps + tap (the OIHW order of a weight gradient) instead of tap * C + c
    split &= 3;
    const int taps = KH * KW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + ty + 8 * r, c = c0 + tx;
        float v = 0.f;
        if (m < M && c < C) {
            const int ox = m % Wo, q = m / Wo, oy = q % Ho, b = q / Ho;
            const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[((size_t)(b * H + iy) * W + ix) * ldx + c];
        }
        tile[ty + 8 * r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, m = m0 + tx;
        if (c < C && m < Mp) {
            const float v = tile[tx][ty + 8 * r];
            const size_t row = chan_major ? (size_t)c * taps + tap : (size_t)tap * C + c;
            if (split == 2) {    // plain bf16 rows
                reinterpret_cast<unsigned short*>(xt)[row * Mp + m] = vidc::bf16_rne(v);
            } else if (split) {  // the 32 pixels of this tile are one K unit of the GEMM: [32 x hi | 32 x lo] in the same 128 bytes
                unsigned short h, l;
                vidc::split_bf16(v, h, l);
                unsigned short* u = reinterpret_cast<unsigned short*>(xt + row * Mp + m0);
                u[tx] = h;
                u[32 + tx] = l;
            } else {
                xt[row * Mp + m] = v;
            }
        }
    }
}
// The same map on 64 pixels x 64 channels per workgroup with 16-byte accesses on both sides (C % 4 == 0, ldx % 4 == 0, 16-byte aligned
// pointers; Mp % 32 == 0 makes every 4-pixel group of a row all-inside or all-outside).  LDS tile [pixel][channel], row pitch 65 floats:
// the 4 x 16 lanes of a wave hit 64 different banks in both phases.
__global__ void __launch_bounds__(256)
im2col_t64_kernel(const float* __restrict__ x, float* __restrict__ xt, int B, int H, int W, int C, int ldx, int Ho, int Wo, int KH, int KW, int stride,
                  int pad, int M, int Mp, int split) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    const int kh = tap / KW, kw = tap - kh * KW;
    const bool chan_major = split & 4;      // rows ordered c * taps + tap instead of tap * C + c (see im2col_t_kernel)
    split &= 3;
    const int taps = KH * KW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + ty + 16 * r, c = c0 + tx * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && c < C) {
            const int ox = m % Wo, q = m / Wo, oy = q % Ho, b = q / Ho;
            const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const float4*>(x + ((size_t)(b * H + iy) * W + ix) * ldx + c);
        }
        float* t = &tile[ty + 16 * r][tx * 4];
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cl = ty + 16 * r, c = c0 + cl, m = m0 + tx * 4;
        if (c < C && m < Mp) {
            const float v[4] = {tile[tx * 4][cl], tile[tx * 4 + 1][cl], tile[tx * 4 + 2][cl], tile[tx * 4 + 3][cl]};
            const size_t ri = chan_major ? (size_t)c * taps + tap : (size_t)tap * C + c;
            float* row = xt + ri * Mp;
            if (split == 2) {    // plain bf16 rows: 4 pixels = 8 bytes
                unsigned short* u = reinterpret_cast<unsigned short*>(xt) + ri * Mp + m;
                *reinterpret_cast<uint2*>(u) = make_uint2((unsigned)vidc::bf16_rne(v[0]) | ((unsigned)vidc::bf16_rne(v[1]) << 16),
                                                          (unsigned)vidc::bf16_rne(v[2]) | ((unsigned)vidc::bf16_rne(v[3]) << 16));
            } else if (split) {  // 4 pixels of one 32-pixel K unit: their hi halves and their lo halves, 8 bytes each
                unsigned short h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) vidc::split_bf16(v[k], h[k], l[k]);
                unsigned short* u = reinterpret_cast<unsigned short*>(row + (m & ~31)) + (m & 31);
                *reinterpret_cast<uint2*>(u) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                *reinterpret_cast<uint2*>(u + 32) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
            } else {
                *reinterpret_cast<float4*>(row + m) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}
// im2col_t64_kernel for the plain-bf16 mode from the bf16 operand copy of x (dense rows [B*H*W][C] of bf16, what the forward conv read)
// instead of the fp32 tensor: half the bytes in, no rounding step (same bits: the copy holds the rounded values).  Rows of xt in channel-major
// order (c * taps + tap), plain bf16, Mp % 64 == 0.
__global__ void __launch_bounds__(256)
im2col_t64_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ xt, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW,
                       int stride, int pad, int M, int Mp) {
    __shared__ unsigned short tile[64][66];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    const int kh = tap / KW, kw = tap - kh * KW, taps = KH * KW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + ty + 16 * r, c = c0 + tx * 4;
        uint2 v = make_uint2(0u, 0u);
        if (m < M && c < C) {
            const int ox = m % Wo, q = m / Wo, oy = q % Ho, b = q / Ho;
            const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const uint2*>(x + ((size_t)(b * H + iy) * W + ix) * C + c);
        }
        unsigned short* t = &tile[ty + 16 * r][tx * 4];
        t[0] = (unsigned short)v.x; t[1] = (unsigned short)(v.x >> 16); t[2] = (unsigned short)v.y; t[3] = (unsigned short)(v.y >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cl = ty + 16 * r, c = c0 + cl, m = m0 + tx * 4;
        if (c < C && m < Mp)
            *reinterpret_cast<uint2*>(xt + ((size_t)c * taps + tap) * Mp + m) =
                make_uint2((unsigned)tile[tx * 4][cl] | ((unsigned)tile[tx * 4 + 1][cl] << 16), (unsigned)tile[tx * 4 + 2][cl] | ((unsigned)tile[tx * 4 + 3][cl] << 16));
    }
}
// Dense bf16 rows x[M][C] -> their transpose xt[C][Mp] (zeros for m >= M; Mp % 64 == 0): the right operand of the weight-gradient GEMM of a
// 1x1 / stride-1 conv in the plain-bf16 mode, from the bf16 operand copy its forward already read (half the bytes of the fp32 source, no
// rounding step: the bits are those vidc_im2col_transposed(split = 2) produces from the fp32 tensor, which rounds the same values).
// 64 x 64 tile through LDS, 16-byte reads along C, 32-byte runs along m out.
__global__ void __launch_bounds__(256)
transpose_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ xt, int M, int C, int Mp) {
    __shared__ unsigned short tile[64][66];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    {
        const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;            // 8 channel-octets x 32 rows, two passes
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ml = ty + 32 * r, m = m0 + ml, c = c0 + tx * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (m < M && c < C) v = *reinterpret_cast<const uint4*>(x + (size_t)m * C + c);
            unsigned short* t = &tile[ml][tx * 8];
            t[0] = (unsigned short)v.x; t[1] = (unsigned short)(v.x >> 16); t[2] = (unsigned short)v.y; t[3] = (unsigned short)(v.y >> 16);
            t[4] = (unsigned short)v.z; t[5] = (unsigned short)(v.z >> 16); t[6] = (unsigned short)v.w; t[7] = (unsigned short)(v.w >> 16);
        }
    }
    __syncthreads();
    const int cl = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16, c = c0 + cl, m = m0 + seg;
    if (c < C && m < Mp) {
        unsigned w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = (unsigned)tile[seg + 2 * k][cl] | ((unsigned)tile[seg + 2 * k + 1][cl] << 16);
        uint4* dst = reinterpret_cast<uint4*>(xt + (size_t)c * Mp + m);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}
// dw_oihw[co][ci][tap] = tmp[co][tap * Cin + ci]
__global__ void __launch_bounds__(TT) wgrad_permute_kernel(const float* __restrict__ tmp, float* __restrict__ dw, int Cout, int Cin, int taps) {
    const long long i = (long long)blockIdx.x * TT + threadIdx.x;
    const long long total = (long long)Cout * Cin * taps;
    if (i >= total) return;
    const int tap = (int)(i % taps);
    const long long q = i / taps;
    const int ci = (int)(q % Cin), co = (int)(q / Cin);
    dw[i] = tmp[((long long)co * taps + tap) * Cin + ci];
}

// stem conv (3x3, stride 2, pad 1, Cin = 1 or 3, NCHW input, no bias): dW[co][ci][kh][kw] = sum dY[m][co] * x[b][ci][2oy+kh-1][2ox+kw-1]
__global__ void __launch_bounds__(TT)
stem_wgrad_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int Cin, int H, int W, int Ho, int Wo, int Cout, int lddy,
                          int rows_per_chunk, double* __restrict__ partial /* [chunk][Cout*Cin*9] */) {
    // One thread per (co, ci) with the nine taps in registers: a row costs one dy load, nine x loads and nine fp64 multiply-adds.  (The
    // first version had one thread per (co, ci, kh, kw): ~20 instructions per multiply-add, 265 M of them -- 300 us at the end of every
    // pyramid's backward.)  Per output the rows are added in the same order as before; a tap outside the image adds dy * 0.
    const int t = blockIdx.x * TT + threadIdx.x;
    if (t >= Cout * Cin) return;
    const int ci = t % Cin, co = t / Cin;
    const long long M = (long long)B * Ho * Wo;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    int ox = (int)(r0 % Wo);                              // (b, oy, ox) of the row walk incrementally: no division in the loop
    const long long q0 = r0 / Wo;
    int oy = (int)(q0 % Ho), b = (int)(q0 / Ho);
    const float* gp = dy + r0 * lddy + co;
    const int n_rows = (int)(r1 - r0);
    const size_t plane_stride = (size_t)Cin * H * W;
    const float* plane = x + ((size_t)b * Cin + ci) * H * W;
    for (int r = 0; r < n_rows; ++r, gp += lddy) {
        const double g = (double)*gp;
        const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + 2 < H && ix0 + 2 < W) {      // wave-uniform: the whole 3x3 patch is inside (all but the border rows):
            const float* p0 = plane + iy0 * W + ix0;                    // nine loads at fixed offsets, no per-tap bounds logic (which was ~200
            const float* p1 = p0 + W;                                   // scalar instructions per row and bound the kernel)
            const float* p2 = p1 + W;
            const float v[9] = {p0[0], p0[1], p0[2], p1[0], p1[1], p1[2], p2[0], p2[1], p2[2]};
#pragma unroll
            for (int k = 0; k < 9; ++k) s[k] += g * (double)v[k];
        } else {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int iy = iy0 + kh;
                const bool row_ok = (unsigned)iy < (unsigned)H;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ix = ix0 + kw;
                    const bool inside = row_ok && (unsigned)ix < (unsigned)W;
                    const float xv = plane[inside ? iy * W + ix : 0];
                    s[kh * 3 + kw] += g * (double)(inside ? xv : 0.f);
                }
            }
        }
        if (++ox == Wo) { ox = 0; if (++oy == Ho) { oy = 0; ++b; plane += plane_stride; } }
    }
    double* out = partial + (size_t)blockIdx.y * ((size_t)Cout * Cin * 9) + (size_t)t * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) out[k] = s[k];
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kStemRows = 64;             // rows per workgroup of vidc_stem_wgrad (M = 153 600 at batch 8: 2400 chunks of 192 threads -- the kernel is
                                          // a chain of dependent L2 latencies per thread, so it wants many short chains rather than few long ones)
constexpr int kRowsPerChunk = 256;        // rows per workgroup of the per-channel reductions: M = 10^4..10^5 rows -> hundreds of workgroups per 64 channels
inline int chunks_for(long long M) { return (int)((M + kRowsPerChunk - 1) / kRowsPerChunk); }
// Rows per workgroup of the per-channel reductions (chan_partial_kernel): about 512 workgroups per launch whatever the shape -- a function of
// (M, C) only, so the summation order of a given tensor never changes.  Multiple of 16 (the row-lanes of a block), 32..256.
inline int rows_for(long long M, int C) {
    const long long cb = (C + 63) / 64;
    long long r = (M * cb + 511) / 512;
    r = (r + 15) / 16 * 16;
    return (int)(r < 32 ? 32 : (r > 256 ? 256 : r));
}
inline int chunks_for(long long M, int C) { const int r = rows_for(M, C); return (int)((M + r - 1) / r); }
inline unsigned blocks(long long n) { return (unsigned)((n + TT - 1) / TT); }
// The final reduction of a BatchNorm's chunk sums inside the consuming kernel's prologue (bn_apply_fold_kernel, bn_bwd_apply_t64_kernel<true>)
// instead of a chan_final launch: every 64 x 64 workgroup re-reads n_chunks x 64 x 2 doubles, so it is only offered where pixel blocks x
// chunks is small -- the ResNet-101 layer-3 / layer-4 maps and the coarse decoder levels, three quarters of a step's BatchNorms.  Same sums,
// same order, same bits (tests/test_training.py).  Round 4 (one thread per channel, 32 dependent loads each): 29.2 against 27.5 ms per
// batch-8 bf16 step.  Round 5 (column_sums: all 256 threads, 16 loads in flight per thread): 26.2-26.3 against 26.15 ms with per-pyramid
// lanes, 27.2 against 27.3 ms grouped -- level, as is finishing the sums by the last workgroup of each channel column of the partial-sum
// kernel (a ticket per column, tried and removed: 26.6-26.7 / 27.8 ms).  A captured step hides the 5 us chan_final launches; nothing is
// gained by removing them (profiles/r5_train_reduction_forms_ab.txt).  Hence OFF by default; the switch (vidc_train_bn_fold,
// VIDC_TRAIN_BN_FOLD=1) stays for measurements.
int g_bn_fold = 0;
inline bool fold_bn(long long M, int C) {
    return g_bn_fold && M < (1ll << 31) && ((M + 63) / 64) * (long long)chunks_for(M, C) <= 4096;
}
}

extern "C" int vidc_train_bn_fold(int enable) {
    const int prev = g_bn_fold;
    if (enable >= 0) g_bn_fold = enable ? 1 : 0;
    return prev;
}

extern "C" size_t vidc_train_scratch_bytes(long long M, int C) {
    return ((size_t)chunks_for(M, C) * 2 * (size_t)C + 2 * (size_t)C) * sizeof(double) + 256;
}

extern "C" int vidc_bn_train_forward_add(const float* x, float* y, long long M, int C, int ldx, int ldy, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float eps, float momentum, int relu, float* save_mean, float* save_rstd,
                                         void* y_bf16, const float* residual, int ldr, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y && gamma && beta && save_mean && save_rstd && scratch, VIDC_ERR_NULL, "vidc_bn_train_forward: null pointer");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || (ldr >= C && ldr % 4 == 0)),
                 VIDC_ERR_SHAPE, "vidc_bn_train_forward: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int nch = chunks_for(M, C);
    double* partial = reinterpret_cast<double*>(scratch);
    double* sums = partial + (size_t)nch * 2 * C;
    hipLaunchKernelGGL(chan_partial_kernel<0>, dim3((C + 63) / 64, nch), dim3(TT), 0, st, x, (const float*)nullptr, (const float*)nullptr, M, C, ldx, 0, 0,
                       (const float*)nullptr, (const float*)nullptr, rows_for(M, C), partial);
    if (fold_bn(M, C)) {
        hipLaunchKernelGGL(bn_apply_fold_kernel, dim3((unsigned)((M + 63) / 64), (C + 63) / 64), dim3(256), 0, st, x, y, (int)M, C, ldx, ldy, partial, nch, eps,
                           momentum, save_mean, save_rstd, running_mean, running_var, gamma, beta, relu, reinterpret_cast<unsigned short*>(y_bf16), residual, ldr);
        VIDC_CHECK_LAUNCH("bn_train_forward (folded)");
        return VIDC_OK;
    }
    hipLaunchKernelGGL(chan_final_kernel<FinalStats>, dim3((C + 7) / 8), dim3(TT), 0, st, partial, nch, C, sums,
                       FinalStats{M, eps, momentum, save_mean, save_rstd, running_mean, running_var});
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks(M * (C / 4))), dim3(TT), 0, st, x, y, M, C, ldx, ldy, save_mean, save_rstd, gamma, beta, relu,
                       reinterpret_cast<unsigned short*>(y_bf16), residual, ldr);
    VIDC_CHECK_LAUNCH("bn_train_forward");
    return VIDC_OK;
}

// The same with the per-channel partial sums ALREADY written by the conv that produced x (VIDC_STATS_OUT: one pair of rows of C doubles per
// block of 32 output rows, in chan_partial_kernel's layout): only the final reduction and the apply pass run here.
extern "C" int vidc_bn_train_forward_stats(const float* x, float* y, long long M, int C, int ldx, int ldy, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float eps, float momentum, int relu, float* save_mean,
                                           float* save_rstd, void* y_bf16, const float* residual, int ldr, const void* conv_stats, void* scratch,
                                           vidc_stream_t stream) {
    VIDC_REQUIRE(x && y && gamma && beta && save_mean && save_rstd && conv_stats && scratch, VIDC_ERR_NULL, "vidc_bn_train_forward_stats: null pointer");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || (ldr >= C && ldr % 4 == 0)),
                 VIDC_ERR_SHAPE, "vidc_bn_train_forward_stats: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int nch = (int)((M + 31) / 32);
    double* sums = reinterpret_cast<double*>(scratch);      // [2][C]
    hipLaunchKernelGGL(chan_final_kernel<FinalStats>, dim3((C + 7) / 8), dim3(TT), 0, st, reinterpret_cast<const double*>(conv_stats), nch, C, sums,
                       FinalStats{M, eps, momentum, save_mean, save_rstd, running_mean, running_var});
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks(M * (C / 4))), dim3(TT), 0, st, x, y, M, C, ldx, ldy, save_mean, save_rstd, gamma, beta, relu,
                       reinterpret_cast<unsigned short*>(y_bf16), residual, ldr);
    VIDC_CHECK_LAUNCH("bn_train_forward_stats");
    return VIDC_OK;
}

extern "C" int vidc_bn_train_forward(const float* x, float* y, long long M, int C, int ldx, int ldy, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float eps, float momentum, int relu, float* save_mean, float* save_rstd,
                                     void* y_bf16, void* scratch, vidc_stream_t stream) {
    return vidc_bn_train_forward_add(x, y, M, C, ldx, ldy, gamma, beta, running_mean, running_var, eps, momentum, relu, save_mean, save_rstd, y_bf16, nullptr, 0,
                                     scratch, stream);
}

extern "C" int vidc_bn_train_backward_t(const float* dy, const float* x, const float* y_relu, float* dx, long long M, int C, int lddy, int ldx, int ldy,
                                        int lddx, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                                        void* dx_bf16, void* dx_bf16_t, int Mp, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(dy && x && gamma && save_mean && save_rstd && dgamma && dbeta && scratch, VIDC_ERR_NULL, "vidc_bn_train_backward: null pointer");
    VIDC_REQUIRE(dx || (dx_bf16 && dx_bf16_t), VIDC_ERR_NULL, "vidc_bn_train_backward: dx may be NULL only when both bf16 forms are written");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!y_relu || ldy % 4 == 0), VIDC_ERR_SHAPE,
                 "vidc_bn_train_backward: bad shape");
    VIDC_REQUIRE(!dx_bf16_t || (Mp >= M && Mp % 64 == 0 && M < (1ll << 31) && (reinterpret_cast<uintptr_t>(dx_bf16_t) & 7) == 0), VIDC_ERR_SHAPE,
                 "vidc_bn_train_backward_t: the transposed copy has rows of Mp = M rounded up to a multiple of 64 pixels, 8-byte aligned");
    hipStream_t st = vidc::as_stream(stream);
    const int nch = chunks_for(M, C);
    double* partial = reinterpret_cast<double*>(scratch);
    double* sums = partial + (size_t)nch * 2 * C;
    hipLaunchKernelGGL(chan_partial_kernel<1>, dim3((C + 63) / 64, nch), dim3(TT), 0, st, dy, x, y_relu, M, C, lddy, ldx, ldy, save_mean, save_rstd, rows_for(M, C), partial);
    if (fold_bn(M, C)) {
        const int mp = dx_bf16_t ? Mp : (int)((M + 63) / 64 * 64);
        hipLaunchKernelGGL(bn_bwd_apply_t64_kernel<true>, dim3(mp / 64, (C + 63) / 64), dim3(256), 0, st, dy, x, y_relu, dx, (int)M, C, lddy, ldx, ldy, lddx,
                           save_mean, save_rstd, gamma, partial, reinterpret_cast<unsigned short*>(dx_bf16), reinterpret_cast<unsigned short*>(dx_bf16_t), mp, nch,
                           dgamma, dbeta);
        VIDC_CHECK_LAUNCH("bn_train_backward (folded)");
        return VIDC_OK;
    }
    hipLaunchKernelGGL(chan_final_kernel<FinalParamGrad>, dim3((C + 7) / 8), dim3(TT), 0, st, partial, nch, C, sums, FinalParamGrad{dgamma, dbeta});
    if (dx_bf16_t)
        hipLaunchKernelGGL(bn_bwd_apply_t64_kernel<false>, dim3(Mp / 64, (C + 63) / 64), dim3(256), 0, st, dy, x, y_relu, dx, (int)M, C, lddy, ldx, ldy, lddx, save_mean,
                           save_rstd, gamma, sums, reinterpret_cast<unsigned short*>(dx_bf16), reinterpret_cast<unsigned short*>(dx_bf16_t), Mp, 0,
                           (float*)nullptr, (float*)nullptr);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks(M * (C / 4))), dim3(TT), 0, st, dy, x, y_relu, dx, M, C, lddy, ldx, ldy, lddx, save_mean, save_rstd,
                           gamma, sums, reinterpret_cast<unsigned short*>(dx_bf16));
    VIDC_CHECK_LAUNCH("bn_train_backward");
    return VIDC_OK;
}

extern "C" int vidc_bn_train_backward(const float* dy, const float* x, const float* y_relu, float* dx, long long M, int C, int lddy, int ldx, int ldy,
                                      int lddx, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                                      void* dx_bf16, void* scratch, vidc_stream_t stream) {
    return vidc_bn_train_backward_t(dy, x, y_relu, dx, M, C, lddy, ldx, ldy, lddx, gamma, save_mean, save_rstd, dgamma, dbeta, dx_bf16, nullptr, 0, scratch, stream);
}

extern "C" int vidc_colsum(const float* dy, long long M, int C, int ld, float* out, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(dy && out && scratch, VIDC_ERR_NULL, "vidc_colsum: null pointer");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0, VIDC_ERR_SHAPE, "vidc_colsum: bad shape (C and ld multiples of 4)");
    hipStream_t st = vidc::as_stream(stream);
    const int nch = chunks_for(M, C);
    double* partial = reinterpret_cast<double*>(scratch);
    double* sums = partial + (size_t)nch * 2 * C;
    hipLaunchKernelGGL(chan_partial_kernel<0>, dim3((C + 63) / 64, nch), dim3(TT), 0, st, dy, (const float*)nullptr, (const float*)nullptr, M, C, ld, 0, 0,
                       (const float*)nullptr, (const float*)nullptr, rows_for(M, C), partial);
    hipLaunchKernelGGL(chan_final_kernel<FinalColsum>, dim3((C + 7) / 8), dim3(TT), 0, st, partial, nch, C, sums, FinalColsum{out});
    VIDC_CHECK_LAUNCH("colsum");
    return VIDC_OK;
}

extern "C" int vidc_add_rows_bf16(const float* a, const float* b, float* y, long long M, int C, int lda, int ldb, int ldy, int relu, void* y_bf16,
                                  vidc_stream_t stream) {
    VIDC_REQUIRE(a && b && y, VIDC_ERR_NULL, "vidc_add_rows: null pointer");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldy % 4 == 0, VIDC_ERR_SHAPE, "vidc_add_rows: bad shape");
    hipLaunchKernelGGL(add_kernel, dim3(blocks(M * (C / 4))), dim3(TT), 0, vidc::as_stream(stream), a, b, y, M, C, lda, ldb, ldy, relu,
                       reinterpret_cast<unsigned short*>(y_bf16));
    VIDC_CHECK_LAUNCH("add_kernel");
    return VIDC_OK;
}

extern "C" int vidc_add_rows(const float* a, const float* b, float* y, long long M, int C, int lda, int ldb, int ldy, int relu, vidc_stream_t stream) {
    return vidc_add_rows_bf16(a, b, y, M, C, lda, ldb, ldy, relu, nullptr, stream);
}

extern "C" int vidc_relu_backward(const float* dy, const float* y, float* dx, long long M, int C, int lddy, int ldy, int lddx, int accumulate,
                                  vidc_stream_t stream) {
    VIDC_REQUIRE(dy && dx, VIDC_ERR_NULL, "vidc_relu_backward: null pointer");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (!y || ldy % 4 == 0), VIDC_ERR_SHAPE, "vidc_relu_backward: bad shape");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks(M * (C / 4))), dim3(TT), 0, vidc::as_stream(stream), dy, y, dx, M, C, lddy, ldy, lddx, accumulate);
    VIDC_CHECK_LAUNCH("relu_bwd_kernel");
    return VIDC_OK;
}

extern "C" int vidc_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int ldx, int lddy, int lddx,
                                          vidc_stream_t stream) {
    VIDC_REQUIRE(x && dy && dx, VIDC_ERR_NULL, "vidc_maxpool3x3s2_backward: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, VIDC_ERR_SHAPE, "vidc_maxpool3x3s2_backward: bad shape");
    const long long Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;      // (the wide kernel indexes x, dx AND dy with 32-bit offsets)
    const bool wide = C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (long long)B * H * W * (ldx > lddx ? ldx : lddx) < (1ll << 31) &&
                      (long long)B * Ho * Wo * lddy < (1ll << 31) &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (wide)
        hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(blocks((long long)B * H * W * (C / 4))), dim3(TT), 0, vidc::as_stream(stream), x, dy, dx, B, H, W, C, ldx, lddy, lddx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks((long long)B * H * W * C)), dim3(TT), 0, vidc::as_stream(stream), x, dy, dx, B, H, W, C, ldx, lddy, lddx);
    VIDC_CHECK_LAUNCH("maxpool_bwd_kernel");
    return VIDC_OK;
}

extern "C" int vidc_upsample_bilinear_ac_backward(const float* dy, float* dx, int B, int h, int w, int C, int lddy, int lddx, int H, int W,
                                                  vidc_stream_t stream) {
    VIDC_REQUIRE(dy && dx, VIDC_ERR_NULL, "vidc_upsample_bilinear_ac_backward: null pointer");
    VIDC_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && H > 0 && W > 0, VIDC_ERR_SHAPE, "vidc_upsample_bilinear_ac_backward: bad shape");
    const bool wide = C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (long long)B * H * W * lddy < (1ll << 31) && (long long)B * h * w * lddx < (1ll << 31) &&
                      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (wide)
        hipLaunchKernelGGL(upsample_bwd4_kernel, dim3(blocks((long long)B * h * w * (C / 4))), dim3(TT), 0, vidc::as_stream(stream), dy, dx, B, h, w, C, lddy, lddx, H, W);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel, dim3(blocks((long long)B * h * w * C)), dim3(TT), 0, vidc::as_stream(stream), dy, dx, B, h, w, C, lddy, lddx, H, W);
    VIDC_CHECK_LAUNCH("upsample_bwd_kernel");
    return VIDC_OK;
}

extern "C" int vidc_head_backward(const float* g_low, const float* x, const float* wgt, float* dx, float* dw, float* dbias, int B, int h, int w, int C,
                                  int ldx, int lddx, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(g_low && x && wgt && dx && dw && dbias && scratch, VIDC_ERR_NULL, "vidc_head_backward: null pointer");
    VIDC_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, VIDC_ERR_SHAPE, "vidc_head_backward: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const long long M = (long long)B * h * w;
    const int nch = chunks_for(M);
    double* partial = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(head_dgrad_kernel, dim3(blocks(M * C)), dim3(TT), 0, st, g_low, wgt, dx, B, h, w, C, lddx);
    hipLaunchKernelGGL(head_wgrad_partial_kernel, dim3(blocks(C), nch), dim3(TT), 0, st, g_low, x, B, h, w, C, ldx, kRowsPerChunk, partial);
    hipLaunchKernelGGL(head_wgrad_final_kernel, dim3((C + TT / kFinalLanes - 1) / (TT / kFinalLanes)), dim3(TT), 0, st, partial, nch, C, dw);
    const long long n = (long long)B * (h + 2) * (w + 2);
    const int nb = (int)((n + (long long)TT * 16 - 1) / ((long long)TT * 16));
    double* p2 = partial + (size_t)nch * C;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(TT), 0, st, g_low, n, 16, p2);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(TT), 0, st, p2, nb, (double*)nullptr, dbias);
    VIDC_CHECK_LAUNCH("head_backward");
    return VIDC_OK;
}

extern "C" size_t vidc_head_backward_scratch_bytes(int B, int h, int w, int C) {
    const long long M = (long long)B * h * w;
    const long long n = (long long)B * (h + 2) * (w + 2);
    return ((size_t)chunks_for(M) * C + (size_t)((n + (long long)TT * 16 - 1) / ((long long)TT * 16)) + 8) * sizeof(double);
}

extern "C" int vidc_masked_l1_loss(const float* pred, const float* gt, long long n, int hw, double* loss, float* dpred, float* terms, void* scratch,
                                   vidc_stream_t stream) {
    VIDC_REQUIRE(pred && gt && loss && dpred && terms && scratch, VIDC_ERR_NULL, "vidc_masked_l1_loss: null pointer");
    VIDC_REQUIRE(n > 0 && hw > 0, VIDC_ERR_SHAPE, "vidc_masked_l1_loss: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    hipLaunchKernelGGL(l1_loss_kernel, dim3(blocks(n)), dim3(TT), 0, st, pred, gt, n, 1.0f / (float)hw, terms, dpred);
    const int nb = (int)((n + (long long)TT * 16 - 1) / ((long long)TT * 16));
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(TT), 0, st, terms, n, 16, reinterpret_cast<double*>(scratch));
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(TT), 0, st, reinterpret_cast<const double*>(scratch), nb, loss, (float*)nullptr);
    VIDC_CHECK_LAUNCH("masked_l1_loss");
    return VIDC_OK;
}

extern "C" int vidc_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, int step,
                              vidc_stream_t stream) {
    VIDC_REQUIRE(p && g && m && v, VIDC_ERR_NULL, "vidc_adam_step: null pointer");
    VIDC_REQUIRE(n > 0 && step >= 1, VIDC_ERR_SHAPE, "vidc_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    VIDC_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                 VIDC_ERR_SHAPE, "vidc_adam_step: the four buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel, dim3(blocks((n + 3) / 4)), dim3(TT), 0, vidc::as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2));
    VIDC_CHECK_LAUNCH("adam_kernel");
    return VIDC_OK;
}

// Gradient buckets in bf16 for the cross-rank SUM (training.GradientBuckets, VIDC_TRAIN_GRAD_BF16=1): 8 values per lane, 32 B in / 16 B out
// (narrow) and back; HBM-bound, one pass each.
__global__ void __launch_bounds__(TT) grad_narrow_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n) {
    const long long i = ((long long)blockIdx.x * TT + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + i + 4);
        *reinterpret_cast<uint4*>(y + i) = make_uint4(vidc::bf16_rne(a.x) | ((unsigned)vidc::bf16_rne(a.y) << 16), vidc::bf16_rne(a.z) | ((unsigned)vidc::bf16_rne(a.w) << 16),
                                                      vidc::bf16_rne(b.x) | ((unsigned)vidc::bf16_rne(b.y) << 16), vidc::bf16_rne(b.z) | ((unsigned)vidc::bf16_rne(b.w) << 16));
    } else {
        for (long long j = i; j < n; ++j) y[j] = vidc::bf16_rne(x[j]);
    }
}

__global__ void __launch_bounds__(TT) grad_widen_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, long long n) {
    const long long i = ((long long)blockIdx.x * TT + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + i);
        *reinterpret_cast<float4*>(y + i) = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xFFFF0000u));
        *reinterpret_cast<float4*>(y + i + 4) = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xFFFF0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xFFFF0000u));
    } else {
        for (long long j = i; j < n; ++j) y[j] = __uint_as_float((unsigned)x[j] << 16);
    }
}

extern "C" int vidc_grad_narrow_bf16(const float* x, void* y_bf16, long long n, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y_bf16, VIDC_ERR_NULL, "vidc_grad_narrow_bf16: null pointer");
    VIDC_REQUIRE(n > 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y_bf16 % 16) == 0, VIDC_ERR_SHAPE, "vidc_grad_narrow_bf16: n > 0 and 16-byte aligned buffers");
    hipLaunchKernelGGL(grad_narrow_kernel, dim3((unsigned)((n + (long long)TT * 8 - 1) / ((long long)TT * 8))), dim3(TT), 0, vidc::as_stream(stream), x,
                       reinterpret_cast<unsigned short*>(y_bf16), n);
    VIDC_CHECK_LAUNCH("grad_narrow_kernel");
    return VIDC_OK;
}

extern "C" int vidc_grad_widen_bf16(const void* x_bf16, float* y, long long n, vidc_stream_t stream) {
    VIDC_REQUIRE(x_bf16 && y, VIDC_ERR_NULL, "vidc_grad_widen_bf16: null pointer");
    VIDC_REQUIRE(n > 0 && ((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)y % 16) == 0, VIDC_ERR_SHAPE, "vidc_grad_widen_bf16: n > 0 and 16-byte aligned buffers");
    hipLaunchKernelGGL(grad_widen_kernel, dim3((unsigned)((n + (long long)TT * 8 - 1) / ((long long)TT * 8))), dim3(TT), 0, vidc::as_stream(stream),
                       reinterpret_cast<const unsigned short*>(x_bf16), y, n);
    VIDC_CHECK_LAUNCH("grad_widen_kernel");
    return VIDC_OK;
}

extern "C" int vidc_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, vidc_stream_t stream) {
    VIDC_REQUIRE(w_oihw && w_packed, VIDC_ERR_NULL, "vidc_pack_conv_weight_dgrad: null pointer");
    VIDC_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cin > 0 && KH > 0 && KW > 0, VIDC_ERR_SHAPE, "vidc_pack_conv_weight_dgrad: Cout must be a multiple of 32");
    const long long total = (long long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_weight_dgrad_kernel, dim3(blocks(total)), dim3(TT), 0, vidc::as_stream(stream), w_oihw, w_packed, Cout, Cin, KH, KW);
    VIDC_CHECK_LAUNCH("pack_weight_dgrad_kernel");
    return VIDC_OK;
}

extern "C" long long vidc_pack_item_blocks(int Cout, int Cin, int KH, int KW, int kind) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || KH * KW > 9 || kind < 0 || kind > 5) return 0;
    const int U = (kind & 4) ? 64 : 32;
    if (((kind & 1) ? Cout : Cin) % U) return 0;          // the K-side channel count must fill whole 128-byte units
    return pack_item_blocks(Cout, Cin, KH, KW, kind);
}

extern "C" int vidc_pack_conv_weights_batched(const vidc_pack_item* items_device, int n_items, long long total_blocks, vidc_stream_t stream) {
    VIDC_REQUIRE(items_device, VIDC_ERR_NULL, "vidc_pack_conv_weights_batched: null pointer");
    VIDC_REQUIRE(n_items > 0 && total_blocks > 0 && total_blocks < (1ll << 31), VIDC_ERR_SHAPE, "vidc_pack_conv_weights_batched: bad item / block count");
    hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)total_blocks), dim3(TT), 0, vidc::as_stream(stream), items_device, n_items);
    VIDC_CHECK_LAUNCH("pack_batched_kernel");
    return VIDC_OK;
}

extern "C" int vidc_cast_bf16(const float* x, void* y, long long rows, int C, int ldx, vidc_stream_t stream) {
    VIDC_REQUIRE(x && y, VIDC_ERR_NULL, "vidc_cast_bf16: null pointer");
    VIDC_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 4 == 0, VIDC_ERR_SHAPE, "vidc_cast_bf16: C must be a multiple of 8");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks(rows * (C / 8))), dim3(TT), 0, vidc::as_stream(stream), x, reinterpret_cast<unsigned short*>(y), rows, C, ldx);
    VIDC_CHECK_LAUNCH("cast_bf16_kernel");
    return VIDC_OK;
}

extern "C" int vidc_zero_stuff(const float* dy, float* z, int B, int Ho, int Wo, int C, int lddy, int stride, int H, int W, vidc_stream_t stream) {
    VIDC_REQUIRE(dy && z, VIDC_ERR_NULL, "vidc_zero_stuff: null pointer");
    VIDC_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && stride >= 1 && (Ho - 1) * stride < H && (Wo - 1) * stride < W,
                 VIDC_ERR_SHAPE, "vidc_zero_stuff: bad shape");
    hipLaunchKernelGGL(zero_stuff_kernel, dim3(blocks((long long)B * H * W * (C / 4))), dim3(TT), 0, vidc::as_stream(stream), dy, z, B, Ho, Wo, C, lddy,
                       stride, H, W);
    VIDC_CHECK_LAUNCH("zero_stuff_kernel");
    return VIDC_OK;
}

namespace {
inline int wgrad_rows_per_chunk(long long M, int Cout, int Cin, int taps) {
    // enough workgroups to fill 256 CUs, chunks of at least 256 pixels (a multiple of 16)
    const long long tiles = (long long)((Cout + 127) / 128) * ((Cin + 127) / 128) * taps;
    long long chunks = (1024 + tiles - 1) / tiles;
    if (chunks < 1) chunks = 1;
    long long rows = (M + chunks - 1) / chunks;
    if (rows < 256) rows = 256;
    if (rows > 1024) rows = 1024;       // short fp32 accumulation chains (the partials are then summed in fp64): keeps dW within ~1e-4 of an
    rows = (rows + 7) / 8 * 8;          // fp64 evaluation even where the sum over 10^4..10^5 pixels cancels heavily
    return (int)rows;
}
}

extern "C" size_t vidc_conv_wgrad_scratch_bytes(int B, int Ho, int Wo, int Cout, int Cin, int KH, int KW) {
    const long long M = (long long)B * Ho * Wo;
    const int rows = wgrad_rows_per_chunk(M, Cout, Cin, KH * KW);
    const long long chunks = (M + rows - 1) / rows;
    return (size_t)chunks * KH * KW * (size_t)Cout * Cin * sizeof(float);
}

extern "C" int vidc_conv_wgrad(const float* dy, const float* x, float* dw_oihw, int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int lddy,
                               int KH, int KW, int stride, int pad, void* scratch, vidc_stream_t stream) {
    VIDC_REQUIRE(dy && x && dw_oihw && scratch, VIDC_ERR_NULL, "vidc_conv_wgrad: null pointer");
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0 && ldx >= Cin && lddy >= Cout,
                 VIDC_ERR_SHAPE, "vidc_conv_wgrad: bad shape");
    VIDC_REQUIRE(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1, VIDC_ERR_SHAPE, "vidc_conv_wgrad: Ho/Wo inconsistent");
    VIDC_REQUIRE((long long)B * H * W * ldx < (1ll << 31) && (long long)B * Ho * Wo * lddy < (1ll << 31), VIDC_ERR_SHAPE,
                 "vidc_conv_wgrad: tensors must stay below 2^31 elements (32-bit offsets)");
    hipStream_t st = vidc::as_stream(stream);
    const long long M = (long long)B * Ho * Wo;
    const int taps = KH * KW;
    const int rows = wgrad_rows_per_chunk(M, Cout, Cin, taps);
    const int chunks = (int)((M + rows - 1) / rows);
    float* partial = reinterpret_cast<float*>(scratch);
    hipLaunchKernelGGL(wgrad_kernel, dim3((Cout + 127) / 128, ((Cin + 127) / 128) * taps, chunks), dim3(256), 0, st, dy, x, B, H, W, Cin, ldx, Ho, Wo, Cout, lddy,
                       KH, KW, stride, pad, rows, partial);
    hipLaunchKernelGGL(wgrad_final_kernel, dim3(blocks((long long)taps * Cout * Cin)), dim3(TT), 0, st, partial, chunks, taps, Cout, Cin, dw_oihw);
    VIDC_CHECK_LAUNCH("conv_wgrad");
    return VIDC_OK;
}

extern "C" int vidc_im2col_transposed(const float* x, float* xt, int B, int H, int W, int C, int ldx, int Ho, int Wo, int KH, int KW, int stride, int pad,
                                      int Mp, int split, vidc_stream_t stream) {
    VIDC_REQUIRE(x && xt, VIDC_ERR_NULL, "vidc_im2col_transposed: null pointer");
    const long long M = (long long)B * Ho * Wo;
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0 && Mp >= M && Mp % 32 == 0 && M < (1ll << 31) &&
                     (long long)KH * KW <= 65535 && split >= 0 && (split & 3) <= 2 && split < 8 && ((split & 3) != 2 || Mp % 64 == 0), VIDC_ERR_SHAPE,
                 "vidc_im2col_transposed: bad shape (Mp = M rounded up to a multiple of 32; 64 for plain bf16 rows)");
    const bool wide = C % 4 == 0 && ldx % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xt)) & 15) == 0;
    if (wide)
        hipLaunchKernelGGL(im2col_t64_kernel, dim3((Mp + 63) / 64, (C + 63) / 64, KH * KW), dim3(256), 0, vidc::as_stream(stream), x, xt, B, H, W, C, ldx, Ho, Wo,
                           KH, KW, stride, pad, (int)M, Mp, split);
    else
        hipLaunchKernelGGL(im2col_t_kernel, dim3(Mp / 32, (C + 31) / 32, KH * KW), dim3(256), 0, vidc::as_stream(stream), x, xt, B, H, W, C, ldx, Ho, Wo, KH, KW,
                           stride, pad, (int)M, Mp, split);
    VIDC_CHECK_LAUNCH("im2col_t_kernel");
    return VIDC_OK;
}

extern "C" int vidc_im2col_transposed_bf16(const void* x_bf16, void* xt_bf16, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                                           int Mp, vidc_stream_t stream) {
    VIDC_REQUIRE(x_bf16 && xt_bf16, VIDC_ERR_NULL, "vidc_im2col_transposed_bf16: null pointer");
    const long long M = (long long)B * Ho * Wo;
    VIDC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0 && Mp >= M && Mp % 64 == 0 && M < (1ll << 31) &&
                     (long long)B * H * W * C < (1ll << 31) && (long long)KH * KW <= 65535 &&
                     ((reinterpret_cast<uintptr_t>(x_bf16) | reinterpret_cast<uintptr_t>(xt_bf16)) & 7) == 0,
                 VIDC_ERR_SHAPE, "vidc_im2col_transposed_bf16: bad shape (C a multiple of 4, Mp = M rounded up to a multiple of 64, 8-byte aligned tensors)");
    hipLaunchKernelGGL(im2col_t64_bf16_kernel, dim3(Mp / 64, (C + 63) / 64, KH * KW), dim3(256), 0, vidc::as_stream(stream),
                       reinterpret_cast<const unsigned short*>(x_bf16), reinterpret_cast<unsigned short*>(xt_bf16), B, H, W, C, Ho, Wo, KH, KW, stride, pad, (int)M, Mp);
    VIDC_CHECK_LAUNCH("im2col_t64_bf16_kernel");
    return VIDC_OK;
}

extern "C" int vidc_transpose_bf16(const void* x_bf16, void* xt_bf16, long long M, int C, int Mp, vidc_stream_t stream) {
    VIDC_REQUIRE(x_bf16 && xt_bf16, VIDC_ERR_NULL, "vidc_transpose_bf16: null pointer");
    VIDC_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && Mp >= M && Mp % 64 == 0 && M < (1ll << 31) &&
                     ((reinterpret_cast<uintptr_t>(x_bf16) | reinterpret_cast<uintptr_t>(xt_bf16)) & 15) == 0,
                 VIDC_ERR_SHAPE, "vidc_transpose_bf16: C a multiple of 8, Mp = M rounded up to a multiple of 64, 16-byte aligned tensors");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Mp / 64, (C + 63) / 64), dim3(256), 0, vidc::as_stream(stream),
                       reinterpret_cast<const unsigned short*>(x_bf16), reinterpret_cast<unsigned short*>(xt_bf16), (int)M, C, Mp);
    VIDC_CHECK_LAUNCH("transpose_bf16_kernel");
    return VIDC_OK;
}

extern "C" int vidc_wgrad_permute(const float* tmp, float* dw_oihw, int Cout, int Cin, int taps, vidc_stream_t stream) {
    VIDC_REQUIRE(tmp && dw_oihw, VIDC_ERR_NULL, "vidc_wgrad_permute: null pointer");
    VIDC_REQUIRE(Cout > 0 && Cin > 0 && taps > 0, VIDC_ERR_SHAPE, "vidc_wgrad_permute: bad shape");
    hipLaunchKernelGGL(wgrad_permute_kernel, dim3(blocks((long long)Cout * Cin * taps)), dim3(TT), 0, vidc::as_stream(stream), tmp, dw_oihw, Cout, Cin, taps);
    VIDC_CHECK_LAUNCH("wgrad_permute_kernel");
    return VIDC_OK;
}

extern "C" int vidc_stem_wgrad(const float* dy, const float* x_nchw, float* dw_oihw, int B, int Cin, int H, int W, int Cout, int lddy, void* scratch,
                               vidc_stream_t stream) {
    VIDC_REQUIRE(dy && x_nchw && dw_oihw && scratch, VIDC_ERR_NULL, "vidc_stem_wgrad: null pointer");
    VIDC_REQUIRE(B > 0 && Cin > 0 && Cin <= 4 && H > 0 && W > 0 && Cout > 0 && lddy >= Cout, VIDC_ERR_SHAPE, "vidc_stem_wgrad: bad shape");
    hipStream_t st = vidc::as_stream(stream);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long M = (long long)B * Ho * Wo;
    const int n = Cout * Cin * 9;
    const int rows = kStemRows;
    const int chunks = (int)((M + rows - 1) / rows);
    double* partial = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(stem_wgrad_partial_kernel, dim3(blocks((long long)Cout * Cin), chunks), dim3(TT), 0, st, dy, x_nchw, B, Cin, H, W, Ho, Wo, Cout, lddy, rows, partial);
    hipLaunchKernelGGL(head_wgrad_final_kernel, dim3((n + TT / kFinalLanes - 1) / (TT / kFinalLanes)), dim3(TT), 0, st, partial, chunks, n, dw_oihw);
    VIDC_CHECK_LAUNCH("stem_wgrad");
    return VIDC_OK;
}

extern "C" size_t vidc_stem_wgrad_scratch_bytes(int B, int Cin, int H, int W, int Cout) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long M = (long long)B * Ho * Wo;
    return (size_t)((M + kStemRows - 1) / kStemRows) * (size_t)Cout * Cin * 9 * sizeof(double);
}
