/*
 * vidc.h -- C ABI of libvidc.so: the MI355X (gfx950) depth-completion inference path.
 *
 * This is the drop-in boundary for the reference's per-frame hot path
 * (MARSLab-UMN/vi_depth_completion, main.py:261-298 `RunDepthCompletion._call_cnn` and the three
 * callables it drives).  The reference has no native code on this path -- it is ~1500 ATen calls per
 * frame -- so every entry point below cites the *Python* interface it replaces (file:line relative to
 * the reference root).  Plain pointers and sizes only; no torch types; no exceptions cross this ABI.
 *
 * Conventions
 *   - all tensors are fp32 device memory unless stated; "NHWC" = [B][H][W][ld] with `ld` >= C the
 *     channel stride (so a tensor can be a channel slice of a wider concat buffer); "NCHW" is dense.
 *   - every launcher takes the hipStream_t to enqueue on, allocates nothing, never synchronises,
 *     and is re-entrant.  Scratch memory is supplied by the caller (`*_workspace_bytes`).
 *   - return value: 0 = ok, <0 = vidc_status; vidc_last_error() gives a thread-local message.
 */
#ifndef VIDC_H
#define VIDC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libvidc.so is built with -fvisibility=hidden: the entry points declared between this push and the pop at the end of the file are the
 * library's ONLY dynamic symbols (tests/test_abi.py compares `nm -D` with this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef void* vidc_stream_t; /* hipStream_t */

enum vidc_status {
    VIDC_OK = 0,
    VIDC_ERR_NULL = -1,      /* null pointer argument          */
    VIDC_ERR_SHAPE = -2,     /* unsupported / inconsistent dims */
    VIDC_ERR_HIP = -3,       /* HIP runtime error               */
    VIDC_ERR_STATE = -4      /* bad handle / call order         */
};

int vidc_version(void);               /* ABI version, currently 1 */
const char* vidc_last_error(void);    /* thread-local, never NULL */
int vidc_device_info(int* n_cu, int* lds_bytes_per_cu, char* arch_name, int arch_name_len);
/* Measurement aid (bench.py; no reference counterpart): VIDC_CLOCK_STAMP_WGS single-thread workgroups (one per XCD of a fresh dispatch)
 * each write out[4 * wg + 0..3] = {XCC id it ran on, shader-clock cycle counter, 100 MHz wall clock, 1} (device buffer of
 * 4 * VIDC_CLOCK_STAMP_WGS int64).  The cycle counters of the XCDs have unrelated offsets, so two stamps are compared per XCC id:
 * (cycles_b - cycles_a) / (ticks_b - ticks_a) x 0.1 = the average shader clock in GHz between them -- what the nominal MFMA peak (quoted
 * at 2.4 GHz) scales by under load. */
#define VIDC_CLOCK_STAMP_WGS 8
int vidc_clock_stamp(long long* out, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 2-DoF gravity-aligned warp        (networks/warping_2dof_alignment.py)
 * ---------------------------------------------------------------------------------------------- */

/* Per-sample geometry record, VIDC_WARP_PARAMS floats per sample:
 *   [0..8] Cg_H_C  [9..17] Cg_R_C  [18..26] Cg_H_C^-1 (= K R^T K^-1)  [27] px_min [28] py_min [29] kw [30] kh [31] 0 */
#define VIDC_WARP_PARAMS 32

/* Replaces _build_homography (:35-58) + the corner-bbox / kw,kh block (:124-140, :229-243).
 * gravity, aligned: [B][3].  K_inv: 9 floats (row-major, the fp32 cast of numpy's fp64 inverse, :15,:20). */
int vidc_warp2dof_params(const float* gravity, const float* aligned, int B, float fx, float fy, float cx, float cy,
                         const float* K_inv, int W, int H, float* params, vidc_stream_t stream);

/* Replaces warp_with_gravity_center_aligned (:108-156): y[b,c,Y,X] = bilinear(x[b,c], H^-1 (X/kw+px_min, Y/kh+py_min, 1)),
 * zero padding; align_corners selects grid_sample's convention (0 = what the oracle/reference do on torch>=1.3).
 * x, y: NCHW [B][C][H][W]. */
int vidc_warp2dof_fwd(const float* x, const float* params, float* y, int B, int C, int H, int W, float cx, float cy,
                      int align_corners, vidc_stream_t stream);

/* Replaces inverse_warp_normal_image_with_gravity_center_aligned (:216-255) fused with the bmm by R^T (:253)
 * and, if normalize != 0, F.normalize(z, dim=1) (networks/surface_normal.py:170).  x, z: NCHW [B][3][H][W]. */
int vidc_warp2dof_inv_rot_norm(const float* x, const float* params, float* z, int B, int H, int W, float cx, float cy,
                               int align_corners, int normalize, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * conv + BN + ReLU stacks           (networks/surface_normal.py:10-145, networks/depth_completion.py:16-147)
 * ---------------------------------------------------------------------------------------------- */

/* OIHW -> packed [Cout][Cin/32][KH][KW][32] (K-contiguous rows for the implicit GEMM; channel-unit major, tap minor so
 * the taps of a unit re-read the same pixels back to back), one group per call.  Cin % 32 == 0. */
int vidc_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, vidc_stream_t stream);

enum vidc_conv_flags {
    VIDC_RELU1 = 1,        /* relu after the first affine                                     */
    VIDC_AFFINE2 = 2,      /* second per-channel affine (bn1_3 -> relu -> bn1, surface_normal.py:41-43) */
    VIDC_RELU2 = 4,        /* relu after the second affine                                    */
    VIDC_RESIDUAL = 8,     /* += residual (Bottleneck identity)                               */
    VIDC_RELU3 = 16,       /* relu after the residual add                                     */
    VIDC_ACCUM = 32,       /* y += result (z1+z2+z3+z4, surface_normal.py:168)                */
    VIDC_SPLIT_OUT = 64,   /* also write the split-bf16 image of the result to y_split (layout of y,
                              same channel stride) so a following bf16x3 conv needs no split pass */
    VIDC_NO_F32_OUT = 128, /* with SPLIT_OUT: skip the fp32 store (nobody reads it)              */
    VIDC_X_PLANAR_GROUPS = 512, /* group g's input is a plane of its own (B*H*W rows of ldx values at x + g*x_gs) instead of a channel slice of
                              rows shared by all groups: the grouped weight-gradient GEMMs of the training step (dY^T of the three pyramids) */
    VIDC_STATS_OUT = 256   /* training, VIDC_PREC_BF16 only, no second affine / residual / accumulate / split output (with `groups` > 1 the rows of
                              the partials hold groups * Cout doubles, group-major like the channels of y):
                              `y_split` points to ceil(M / 32) x 2 x Cout doubles and receives, per block of 32 output rows, the
                              per-channel sum and sum of squares of the fp32 result -- the partials vidc_bn_train_forward_stats
                              reduces, so the train-mode BatchNorm behind the conv needs no pass of its own over the tensor */
};

/* Fused nn.Conv2d(+bias) -> BatchNorm2d(eval) -> ReLU [-> BatchNorm2d -> ReLU] [+ identity -> ReLU].
 * Implicit GEMM on fp32 MFMA: M = B*Ho*Wo, N = Cout, K = KH*KW*Cin.  Cin % 32 == 0, Cout % 32 == 0.
 * `groups` independent convolutions of identical shape run in one launch (the three ResNet-101 pyramids of
 * ModifiedFPN, depth_completion.py:155-157); group g uses x + g*x_gs, w + g*w_gs, y + g*y_gs, ... */
typedef struct vidc_conv_desc {
    const float* x;        /* NHWC input,  channel stride ldx                         */
    const float* w;        /* packed weights [Cout][KH*KW*Cin] (vidc_pack_conv_weight) */
    float* y;              /* NHWC output, channel stride ldy                         */
    const float* scale1;   /* [Cout] gamma/sqrt(var+eps)            (1 if no BN)      */
    const float* shift1;   /* [Cout] beta - mean*scale + bias*scale (bias if no BN)   */
    const float* scale2;   /* [Cout] or NULL                                          */
    const float* shift2;
    const float* residual; /* NHWC, channel stride ldr, or NULL                       */
    float* workspace;      /* split-K scratch, or NULL: VIDC_SPLITK_COUNTERS ticket counters (uint32, MUST be zero before the first
                              launch that uses the buffer; every launch leaves them zero) followed by splitk*groups*M*Cout floats
                              of partials.  Launches that share a workspace must be ordered (same stream).             */
    int32_t B, H, W, Cin, ldx;
    int32_t Ho, Wo, Cout, ldy, ldr;
    int32_t KH, KW, stride, pad;
    int32_t flags;         /* vidc_conv_flags                                         */
    int32_t groups;
    int64_t x_gs, w_gs, y_gs, r_gs, p_gs;   /* per-group element strides (p_gs: scale/shift) */
    int32_t tile;          /* vidc_conv_tile, or VIDC_TILE_AUTO                       */
    int32_t splitk;        /* >= 1                                                    */
    int32_t precision;     /* vidc_conv_precision: with VIDC_PREC_BF16X3, x and w are the
                              hi|lo bf16 images made by vidc_split_bf16x3 /
                              vidc_pack_conv_weight_bf16x3 (same strides as fp32)     */
    int32_t dilation;      /* tap spacing of the kernel (nn.Conv2d dilation); 0 or 1 = dense                  */
    void* y_split;         /* split-bf16 image of y (VIDC_SPLIT_OUT), the channel-sum partials (VIDC_STATS_OUT), or NULL */
} vidc_conv_desc;

/* Workgroup tilings (BM x BN output tile; _Kn = n k-slices reduced inside the workgroup through LDS). */
enum vidc_conv_tile { VIDC_TILE_AUTO = 0, VIDC_TILE_128x128 = 1, VIDC_TILE_128x64 = 2, VIDC_TILE_64x128 = 3,
                      VIDC_TILE_64x64 = 4, VIDC_TILE_64x64_K2 = 5, VIDC_TILE_32x64_K2 = 6, VIDC_TILE_32x32_K4 = 7,
                      VIDC_TILE_32x128 = 8, VIDC_TILE_32x32_K8 = 9,
                      /* same tilings with deeper LDS rings (_Dn = n stages) */
                      VIDC_TILE_32x64_K2_D5 = 10, VIDC_TILE_32x32_K4_D4 = 11, VIDC_TILE_32x128_D6 = 12, VIDC_TILE_64x64_K2_D4 = 13,
                      /* _L: same tile with loader waves -- as many extra waves as compute waves issue all LDS-DMA */
                      VIDC_TILE_32x64_K2_L = 14, VIDC_TILE_32x64_K2_D5_L = 15, VIDC_TILE_32x32_K4_D4_L = 16, VIDC_TILE_64x64_L = 17,
                      VIDC_TILE_64x64_K2_D4_L = 18, VIDC_TILE_64x128_L = 19, VIDC_TILE_128x64_L = 20,
                      VIDC_TILE_64x32_K2 = 21, VIDC_TILE_64x32_K2_D5 = 22, VIDC_TILE_64x32_K2_D5_L = 23,
                      /* 64x64 wave tiles (fewer LDS bytes per MFMA), 3-deep ring, with and without loader waves */
                      VIDC_TILE_128x128_D3 = 24, VIDC_TILE_128x128_D3_L = 25,
                      /* 8 waves of 64x64: 128 FLOP per ingested byte (the large layers are bound by the L2 -> LDS DMA rate) */
                      VIDC_TILE_256x128 = 26, VIDC_TILE_128x256 = 27,
                      /* _D2: 2-deep rings, 32-48 KB of LDS per workgroup -- three or more workgroups (of this launch or of another
                       * stream's) per CU: latency is covered by co-resident workgroups instead of a deep ring per workgroup */
                      VIDC_TILE_32x64_K2_D2 = 28, VIDC_TILE_64x64_D2 = 29, VIDC_TILE_32x32_K4_D2 = 30, VIDC_TILE_64x128_D2 = 31,
                      VIDC_TILE_64x32_K2_D2 = 32,
                      /* _P: loader waves + pipelined fragment reads -- every ds_read of the compute waves is issued behind an MFMA, the
                       * first k-half of the next stage across the stage barrier (bf16x3 / bf16; fp32 operands run the _L form) */
                      VIDC_TILE_128x128_D4_P = 33, VIDC_TILE_128x128_D3_P = 34, VIDC_TILE_64x64_D4_P = 35, VIDC_TILE_128x64_D4_P = 36,
                      VIDC_TILE_64x64_K2_D4_P = 37, VIDC_TILE_64x32_K2_D5_P = 38, VIDC_TILE_32x64_K2_D5_P = 39,
                      /* grouped GEMMs with at most 96 rows (the Winograd-domain products of the small maps): fp32, 1x1 / stride 1, flags <= RELU1,
                         p_gs == 0, Cin % 64 == 0, Cin >= 128.  One workgroup streams a chunk of consecutive groups of one 32-channel n-tile through a
                         continuous LDS ring (csrc/wgemm.hip); `splitk` = number of group chunks (<= 1: automatic), no workspace. */
                      VIDC_TILE_G96x32_STREAM = 40,
                      /* the same with 64-channel n-tiles, 12 waves and a 3-deep ring: one workgroup per CU with two stages (80 KB) of loads in flight */
                      VIDC_TILE_G96x64_STREAM3 = 41,
                      /* Winograd F(4x4, 3x3) in one launch (csrc/wfused.hip): the descriptor of a 3x3 / stride 1 / pad 1 conv in fp32, flags within
                         RELU1 | AFFINE2 | RELU2, Cin % 32 == 0 -- except that `w` holds U = G g G^T of vidc_winograd_weight_transform(m = 4) re-ordered by
                         vidc_winograd_weight_pack_fused, 36 * Cout * Cin floats per group (= w_gs).  Input transform, the 36 products and the output transform + epilogue of a
                         block of 16 tiles x 32 output channels run in one workgroup: no V / M tensors, no workspace.  For the small maps
                         (a workgroup multiplies for 36 * Cin / 32 stages: a few hundred workgroups should cover the layer). */
                      VIDC_TILE_WINO4_FUSED = 42,
                      VIDC_TILE_COUNT = 43 };

/* Arithmetic of the contraction.  FP32: v_mfma_f32_32x32x2_f32 on fp32 operands (exact fp32, the reference mode).
 * BF16X3: every operand is split as x = hi + lo (bf16 each, round-to-nearest-even) and each product is computed as
 * lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative per product; whole-path
 * depth RMSE 1.4e-5 against fp32, bar 1e-3).  A 32-channel unit of a split tensor is [32 x bf16 hi | 32 x bf16 lo],
 * i.e. the same 128 bytes as 32 floats, so all strides/offsets of the fp32 layout carry over unchanged.
 * BF16 (the arithmetic BASELINE configs[4] names for training): plain bf16 operands (vidc_cast_bf16 / pack kinds 4, 5), fp32
 * accumulation, fp32 output.  128 bytes of a row are 64 bf16 channels, so the descriptor counts TWO channels per element:
 * Cin, ldx, x_gs = bf16 channels / 2 (Cin a multiple of 32, i.e. 64 bf16 channels), w_gs = Cout*KH*KW*Cin.  2 MFMAs per 16 k. */
enum vidc_conv_precision { VIDC_PREC_FP32 = 0, VIDC_PREC_BF16X3 = 1, VIDC_PREC_BF16 = 2 };
/* fp32 NHWC rows [rows][ldx] (first C channels) -> dense bf16 rows [rows][C] (round to nearest even); C % 8 == 0. */
int vidc_cast_bf16(const float* x, void* y, long long rows, int C, int ldx, vidc_stream_t stream);
/* fp32 NHWC rows [rows][ldx] (first C channels) -> split image [rows][C/32][hi|lo]; C % 32 == 0. */
int vidc_split_bf16x3(const float* x, void* y, long long rows, int C, int ldx, vidc_stream_t stream);
/* OIHW fp32 -> split packed weights [Cout][Cin/32][KH][KW][hi|lo] (one group per call). */
int vidc_pack_conv_weight_bf16x3(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, vidc_stream_t stream);

int vidc_conv2d_bn_act(const vidc_conv_desc* d, vidc_stream_t stream);
#define VIDC_SPLITK_COUNTERS 16384   /* ticket counters at the head of a split-K workspace (one per output tile) */
size_t vidc_conv2d_workspace_bytes(const vidc_conv_desc* d);
/* Fills d->tile / d->splitk with the heuristic choice for this shape on the current device. */
int vidc_conv2d_plan(vidc_conv_desc* d);

/* ---- Winograd F(m x m, 3x3), m = 2 or 4, for the 3x3 / stride 1 / pad 1 Conv2d + BatchNorm2d + ReLU layers (the decoders'
 * `nn.Conv2d(c, c, 3, 1, 1)`: surface_normal.py:75,84,96,104,116,124,132,141; depth_completion.py:77,86,98,106,118,126,134,143) --------
 * conv = output_transform( GEMM( input_transform(x), weight_transform(w) ) ) with a = m + 2:
 *   V, M : [tile][gg][pos][Cin | Cout] rows of a*a*C values (C = G * Cin resp. G * Cout; group gg of the NHWC tensor sits at channel
 *          gg * Cin), tile = (b * th + ty) * tw + tx, th = ceil(H / m), tw = ceil(W / m) (vidc_winograd_tiles), pos = xi * a + nu;
 *   U    : [pos][Cout][Cin] per group.
 * The GEMM is vidc_conv2d_bn_act as a 1x1 conv over B = 1, H = 1, W = tiles with groups = G * a*a, x_gs = Cin, w_gs = Cout * Cin,
 * y_gs = Cout, ldx = a*a*G*Cin, ldy = a*a*G*Cout and an identity epilogue (scale 1, shift 0, flags 0); 4 (m = 2) resp. 2.25 (m = 4)
 * multiply-adds per output and input channel instead of 9.  fp32 throughout (U in fp64, rounded once): the oracle's whole-frame depth
 * moves by RMSE 1.0e-6 / 1.3e-6 against the direct form (bar 1e-3). */
int vidc_winograd_tiles(int H, int W, int m, int* th, int* tw);
/* OIHW [Cout][Cin][3][3] -> U [a*a][Cout][Cin] = G g G^T (one group per call). */
int vidc_winograd_weight_transform(const float* w_oihw, float* u, int Cout, int Cin, int m, vidc_stream_t stream);
/* x NHWC [B][H][W][ldx] (C = G * Cin channels) -> V = B^T d B, zero padding 1.  split != 0: V is written as the split-bf16 image
 * (the operand format of VIDC_PREC_BF16X3; Cin % 32 == 0) instead of fp32.  ldv: values per row of V (0 = dense, a*a*C; larger when
 * x / v address a contiguous range of the groups of a wider tensor). */
/* U of vidc_winograd_weight_transform(m = 4) ([36][Cout][Cin], one group) -> the order the fused kernel (VIDC_TILE_WINO4_FUSED) streams it in:
 * [36][Cout / 32][Cin / 32][2][2][64 lanes][4] with lane = 16 kq + n holding U[pos][32 nb + 16 nblk + n][32 kc + 8 kq + 4 h + 0..3] -- every load
 * instruction of a product wave is 1 KiB contiguous.  Same size; not in place.  Cout % 32 == 0, Cin % 32 == 0. */
int vidc_winograd_weight_pack_fused(const float* u, float* u_packed, int Cout, int Cin, vidc_stream_t stream);
int vidc_winograd_input_transform(const float* x, void* v, int B, int H, int W, int C, int ldx, int Cin, int m, int split, int ldv,
                                  vidc_stream_t stream);
/* M -> y NHWC [B][Ho][Wo][ldy] = epilogue(A^T M A): flags = VIDC_RELU1 | VIDC_AFFINE2 | VIDC_RELU2 | VIDC_SPLIT_OUT | VIDC_NO_F32_OUT with the
 * meaning they have in vidc_conv_desc; scale / shift: [C] (group-major, as the conv's); ldm: values per row of M (0 = dense). */
int vidc_winograd_output_transform(const float* mm, float* y, void* y_split, const float* scale1, const float* shift1,
                                   const float* scale2, const float* shift2, int B, int Ho, int Wo, int C, int Cout, int ldy, int m,
                                   int flags, int ldm, vidc_stream_t stream);

/* Stem conv 3x3 stride 2 pad 1, Cin in {1,3}, no BN, optional ReLU (conv1_1, surface_normal.py:17-18).
 * x: NCHW [B][Cin][H][W] -> y: NHWC [B][Ho][Wo][ldy].  w: OIHW as in the checkpoint. */
int vidc_stem_conv3x3s2(const float* x, const float* w_oihw, float* y, int B, int Cin, int H, int W, int Cout, int ldy,
                        int relu, void* y_split, int split_ch0, vidc_stream_t stream);

/* The same conv reading its input THROUGH the gravity-aligned forward warp (warp_with_gravity_center_aligned, :108-156, in front of
 * `self.resnet_pyramids(warped)`, surface_normal.py:163-164): the patch loader gathers x with the tap sets of vidc_warp2dof_fwd (same code,
 * same bits), the warped image is never stored.  x: NCHW [B][3][H][W]; warp_params: the records of vidc_warp2dof_params. */
int vidc_stem_conv3x3s2_warped(const float* x, const float* warp_params, const float* w_oihw, float* y, int B, int H, int W, int Cout, int ldy,
                               int relu, void* y_split, int split_ch0, float cx, float cy, int align_corners, vidc_stream_t stream);

/* nn.MaxPool2d(3, 2, 1) on NHWC (surface_normal.py:44). */
int vidc_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, void* y_split, vidc_stream_t stream);

enum vidc_up_flags { VIDC_UP_RELU = 1, VIDC_UP_ACCUM = 2, VIDC_UP_NO_F32_OUT = 4,
                     VIDC_UP_SUM_GROUPS_SHIFT = 8 /* bits 8..15 = G > 1: x holds G groups of C channels (group g at channel g*C); their
                                                      upsampled (and ReLU'd) values are summed into the C output channels, g = 0 first */ };
/* nn.UpsamplingBilinear2d(size) == bilinear, align_corners=True, on NHWC (surface_normal.py:88 ...). */
int vidc_upsample_bilinear_ac(const float* x, float* y, int B, int h, int w, int C, int ldx, int H, int W, int ldy,
                              int flags, void* y_split, vidc_stream_t stream);
/* y_split (stem / maxpool / upsample; may be NULL): additionally write the result as the split-bf16 image that the bf16x3
 * convs read (layout: vidc_split_bf16x3), with the same row stride ldy (a multiple of 32) -- saves the separate split
 * launch.  maxpool / stem: y may then be NULL (no fp32 copy); upsample: VIDC_UP_NO_F32_OUT.  stem: y_split is the image of
 * the whole [.., ldy] tensor and split_ch0 the first channel this launch writes (y itself points at that channel). */

/* nn.AvgPool2d((kh,kw), stride=(sh,sw), padding=(ph,pw)), count_include_pad=True, on NHWC (FullImageEncoder.global_pooling,
 * networks/surface_normal_dorn.py:10).  Ho = (H + 2ph - kh) / sh + 1, Wo likewise. */
int vidc_avgpool2d(const float* x, float* y, int B, int H, int W, int C, int ldx, int kh, int kw, int sh, int sw, int ph, int pw, int ldy,
                   vidc_stream_t stream);

/* torch.nn.functional.normalize(x, dim=1) on NCHW [B][C][HW] (networks/surface_normal_dorn.py:154). */
int vidc_normalize_nchw(const float* x, float* y, int B, int C, int HW, vidc_stream_t stream);

/* Prediction head tail: 1x1 conv Cin -> Cout (Cout <= 4) with zero padding `pad` (the reference's
 * Conv2d(192,1,1,1,1), depth_completion.py:144, pads a 1x1 conv -> 62x82 map whose border equals the bias),
 * then UpsamplingBilinear2d(size=(H,W)) and optional ReLU.  x: NHWC [B][h][w][ldx]; lowres: scratch NCHW
 * [B][Cout][h+2pad][w+2pad]; y: NCHW [B][Cout][H][W]. */
int vidc_head_conv1x1_upsample(const float* x, const float* w, const float* bias, float* lowres, float* y, int B, int h,
                               int w_in, int Cin, int ldx, int Cout, int pad, int H, int W, int relu, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * plane block                        (main.py:29-190, 277-297)
 * ---------------------------------------------------------------------------------------------- */

#define VIDC_MAX_HYP 300          /* num_hypotheses, main.py:38,68 */
#define VIDC_PLANE_RECORD 16
/* A plane "slot" = one plane id of one image of the batch: int32[4] = {image index b, plane id, offset into hyp_pix,
 * n_hyp}.  Every stage below is ONE launch over all slots (the reference loops in Python, main.py:147-182,279-283).
 * Per-slot result record (VIDC_PLANE_RECORD floats):
 *   [0..2] n_bar  [3] plane offset d  [4] n_inliers (normal RANSAC)  [5] mean |angle| in degrees  [6] accepted (<= 20 deg)
 *   [7] n_pts (sparse depths on the inliers)  [8] mean_depth  [9] n_offset_inliers
 *   [10] 1 = depth written, 0 = plane failed the validity tests, -1 = n_pts > VIDC_MAX_HYP and no host draw was supplied
 *        for this slot (main.py:75-78; see vidc_plane_offset_dense)  [11] n projected pixels  [12] best hypothesis row */

/* Bytes of device scratch the three plane stages below share for one batch (chunk partial sums, validity statistics,
 * the per-image row-major list of sparse-depth pixels). */
size_t vidc_plane_scratch_bytes(int n_slots, int B, int HW);

/* mean_normal_ranasc (main.py:38-62) + the inlier write-back of main.py:157.
 * normals: NCHW [B][3][HW] unit normals; ids: [B][HW] uint8 plane-id maps; hyp_pix: flat pixel indices of the
 * hypothesis normals of every slot, concatenated (the host draws np.random.permutation exactly like main.py:43);
 * inlier_mask: [n_slots][HW] uint8 out; counts: [n_slots][VIDC_MAX_HYP] int32 out (inliers per hypothesis);
 * the mean normal / mean angle partial sums stay in `scratch` for vidc_plane_offset. */
int vidc_plane_ransac_normal(const float* normals, const uint8_t* ids, const int32_t* slots, int n_slots,
                             const int32_t* hyp_pix, int HW, uint8_t* inlier_mask, int32_t* counts, void* scratch,
                             vidc_stream_t stream);

/* Finishes the record of every slot (n_bar, mean |angle|, the 20-degree acceptance test of main.py:162), then
 * plane_offset_ransac (main.py:68-101) with the point cloud of main.py:171-173: points homo*depth on
 * inlier_mask & depth>0 (row-major order), hypotheses d_j = -n.P_j (all points, n_pts <= VIDC_MAX_HYP, :76-77),
 * inliers |d_j + n.P| < 0.1, d = -mean(n.P over the best hypothesis' inliers).  homo: [B][HW][3]; depth: [B][HW]. */
int vidc_plane_offset(const float* homo, const float* depth, const int32_t* slots, int n_slots, int B,
                      const uint8_t* inlier_mask, const int32_t* counts, int HW, void* scratch, float* records,
                      vidc_stream_t stream);

/* The same, with the subsampled branch of plane_offset_ransac (main.py:75-78: more than 300 points on a plane ->
 * hypotheses = np.random.permutation(np.r_[0:n_pts])[0:300]).  The permutation is a HOST draw from numpy's legacy generator in
 * the reference's draw order, so the protocol has two passes: the first call (dense_* NULL, or dense_n[slot] != n_pts) flags
 * such slots (record[10] = -1, record[7] = n_pts); the host draws and calls again with dense_hyp [n_slots][VIDC_MAX_HYP] = the
 * 300 ranks (into the row-major list of the plane's points), dense_n [n_slots] = the n_pts each row was drawn for (0 = none),
 * dense_dots [n_slots][HW] floats of scratch.  All three NULL = vidc_plane_offset. */
int vidc_plane_offset_dense(const float* homo, const float* depth, const int32_t* slots, int n_slots, int B,
                            const uint8_t* inlier_mask, const int32_t* counts, int HW, void* scratch, float* records,
                            const int32_t* dense_hyp, const int32_t* dense_n, float* dense_dots, vidc_stream_t stream);

/* generate_depth_from_plane (main.py:110-127) -- the normal->depth plane projection: depth = -d/(n.homo) on
 * mask & |n.homo| > 1e-3; the plane is dropped when > 5% of its values exceed 10*mean_depth, any exceeds 10 m or any
 * is negative; otherwise the values are written into plane_depth [B][HW] (pre-initialised with the sparse depth). */
int vidc_plane_project_depth(const float* homo, const int32_t* slots, int n_slots, const uint8_t* inlier_mask, int HW,
                             void* scratch, float* records, float* plane_depth, vidc_stream_t stream);

/* main.py:186-187 (the original sparse depths override the plane depths) fused with what the enrichment needs from the
 * device (main.py:287-289): plane_depth[p] = depth[p] where depth[p] > 0, and `info` (vidc_plane_info_count(B,HW)
 * int32) = for every image the number of pixels with plane_depth > 0 in each 256-pixel chunk, followed by one word:
 * the number of slots whose record is flagged -1.  This buffer is the ONLY device->host read of the whole path. */
int vidc_plane_info_count(int B, int HW);
int vidc_plane_finalize(const float* depth, float* plane_depth, int B, int HW, const float* records, int n_slots,
                        int32_t* info, vidc_stream_t stream);

/* extract_plane_images_from_normal_image (main.py:130-190) for a whole batch as ONE call: plane_depth <- depth, then
 * vidc_plane_ransac_normal, vidc_plane_offset (vidc_plane_offset_dense when any dense_* is given), vidc_plane_project_depth and
 * vidc_plane_finalize with these arguments, in that order on `stream` (n_slots == 0: copy + finalize).  Same kernels, same results;
 * the stream modes use it so that an item's launch-bound plane kernels are enqueued back to back. */
int vidc_plane_block(const float* normals, const uint8_t* ids, const int32_t* slots, int n_slots, const int32_t* hyp_pix, int B, int HW,
                     const float* homo, const float* depth, uint8_t* inlier_mask, int32_t* counts, void* scratch, float* records,
                     const int32_t* dense_hyp, const int32_t* dense_n, float* dense_dots, float* plane_depth, int32_t* info,
                     vidc_stream_t stream);

/* main.py:290-294: for image b the sub[k]-th nonzeros (row-major, like torch.nonzero) of plane_depth[b] are copied
 * into enriched[b] (a clone of the sparse depth).  sub: sorted unique indices drawn on the host with
 * np.unique(np.random.randint(...)); image b owns sub[sub_offsets[b] .. sub_offsets[b+1]); chunk_base: [B][chunks]
 * exclusive prefix sums of the chunk counts returned by vidc_plane_finalize. */
int vidc_enrich_scatter(const float* plane_depth, const int32_t* sub, const int32_t* sub_offsets, const int32_t* chunk_base,
                        int B, int HW, float* enriched, vidc_stream_t stream);

/* The same with main.py:286 (`depth_enriched = ds.clone()`) fused in: EVERY pixel of enriched [B][HW] is written (the sparse depth,
 * or the selected plane depth), so the caller neither clones nor initialises it; sub_offsets[b+1] == sub_offsets[b] gives a plain copy. */
int vidc_enrich_scatter_from(const float* plane_depth, const float* sparse_depth, const int32_t* sub, const int32_t* sub_offsets,
                             const int32_t* chunk_base, int B, int HW, float* enriched, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * frame pre-processing on the device   (dataset.py:461-510)
 * ---------------------------------------------------------------------------------------------- */

/* HOST function (no GPU): the coefficient tables of Pillow's Image.resize(..., Image.BILINEAR) for one axis (Resample.c
 * precompute_coeffs + normalize_coeffs_8bpc): bounds[out][2] = (first input index, tap count), coeffs[out][*ksize_out] 22-bit
 * fixed point.  Call with bounds = coeffs = NULL to query *ksize_out. */
int vidc_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int coeffs_capacity, int* ksize_out);

/* color_img.resize((Wo, Ho), resample=Image.BILINEAR) followed by transforms.ToTensor() (dataset.py:470-471), bit-identical on
 * 8-bit images: src uint8 [B][H][W][C] (device) -> dst float [B][C][Ho][Wo] in [0,1].  Tables from vidc_resize_coeffs, on device. */
int vidc_resize_bilinear_u8_to_chw(const uint8_t* src_hwc, float* dst_chw, int B, int H, int W, int C, int Ho, int Wo,
                                   const int32_t* bounds_x, const int32_t* coeffs_x, int ksize_x, const int32_t* bounds_y,
                                   const int32_t* coeffs_y, int ksize_y, vidc_stream_t stream);

/* The sparse-depth map of dataset.py:495-510: tracks double [N][4] = (id, X, Y, Z) for all images (image b owns rows
 * offsets[b] .. offsets[b+1]); col = int(fx*X/Z + cx), row = int(fy*Y/Z + cy) in float64, depth[row][col] = Z, a later track
 * overwrites an earlier one.  depth float [B][H][W] is zero-filled by the call. */
int vidc_rasterize_sparse_depth(const double* tracks, const int32_t* offsets, int B, double fx, double fy, double cx, double cy,
                                float* depth, int H, int W, vidc_stream_t stream);
/* Ground-truth depth of the training / evaluation streams: KinectAzureDataset / ScanNet loaders, dataset.py:283-286 --
 *   Image.open(depth).convert('F').resize((320, 240), resample=Image.NEAREST); torch.Tensor(np.array(.)) / 1000.0
 * vidc_nearest_table (host, no GPU): source index per output coordinate of Pillow's NEAREST resize along one axis (Geometry.c
 * ImagingScaleAffine).  vidc_resize_nearest_u16_depth: src uint16 [B][H][W] millimetres (device) -> dst float [B][Ho][Wo] = src / divisor
 * (1000: metres), tables on the device.  Bit-identical to the Pillow + torch sequence above. */
int vidc_nearest_table(int in_size, int out_size, int32_t* table);
int vidc_resize_nearest_u16_depth(const uint16_t* src, float* dst, int B, int H, int W, int Ho, int Wo, const int32_t* xtab,
                                  const int32_t* ytab, float divisor, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * evaluation statistics and output conversion on the device   (network_run.py:42-50, 72-82, 198-225, 387-403)
 * ---------------------------------------------------------------------------------------------- */

/* Sufficient statistics of the reference's DEPTH ERROR STATS over the pixels with gt > 0:
 * stats[0] = n valid, [1] = sum |gt - pred|, [2] = sum (gt - pred)^2, [3..7] = #{max(gt/pred, pred/gt) < 1.05, 1.10, 1.25, 1.25^2,
 * 1.25^3}; fp64, fixed summation order.  accumulate != 0: added to what stats holds (running totals over a test set).
 * scratch: vidc_depth_metrics_scratch_bytes(n) bytes. */
size_t vidc_depth_metrics_scratch_bytes(long long n);
int vidc_depth_metrics(const float* pred, const float* gt, long long n, double* stats, int accumulate, void* scratch, vidc_stream_t stream);

/* SaveDepthsToImage's pixel conversion: (depths * 1000).astype(np.uint32). */
int vidc_depth_to_mm_u32(const float* depth, uint32_t* mm, long long n, vidc_stream_t stream);

/* NORMAL ERROR STATS (network_run.py:204-214, 389-397).  pred, gt: [B][3][HW] fp32 (normalised here like the reference does),
 * mask: [B][HW] fp32, valid where > 0.  err [B*HW] receives every pixel's angle error in degrees (fp32; all-ones bits where the mask is
 * off); stats[8] (fp64) (+)= n, sum e, sum e^2, counts of e < 5, 7.5, 11.25, 22.5, 30.  scratch: vidc_depth_metrics_scratch_bytes(B*HW). */
int vidc_normal_metrics(const float* pred, const float* gt, const float* mask, int B, int HW, float* err, double* stats,
                        int accumulate, void* scratch, vidc_stream_t stream);

/* One 16-bit radix digit of the bit patterns of n non-negative floats, counted into hist[65536] (uint32, accumulated: zero it first;
 * all-ones entries are skipped).  hi_filter < 0: the upper 16 bits; otherwise the lower 16 bits of the values whose upper 16 bits equal
 * hi_filter.  Two passes select any order statistic (the reference's np.median) exactly; the counts sum over batches and ranks. */
int vidc_hist_u16(const float* vals, long long n, int hi_filter, uint32_t* hist, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * plane-mask head: the two native kernels it reaches   (plane_mask_detection/maskrcnn_benchmark/csrc)
 * ---------------------------------------------------------------------------------------------- */

/* Greedy non-maximum suppression (csrc/cuda/nms.cu:23-131 incl. its host loop; csrc/cpu/nms_cpu.cpp:5-67), entirely on the device.
 * boxes_xyxy float [n][4] (16-byte aligned), "+1" area convention; order int32 [n] = box indices by descending score (the reference
 * sorts with scores.sort(0, descending=True)); a box is suppressed by a kept, higher-scored one when IoU > threshold (inclusive = 0,
 * nms.cu:57) or IoU >= threshold (inclusive = 1, nms_cpu.cpp:60).  keep int32 [n]: the surviving ORIGINAL indices in ascending
 * order (like the reference's sorted result), *n_keep their count (device memory).  scratch: vidc_nms_scratch_bytes(n). */
size_t vidc_nms_scratch_bytes(int n);
int vidc_nms(const float* boxes_xyxy, const int32_t* order, int n, float threshold, int inclusive, int32_t* keep, int32_t* n_keep,
             void* scratch, vidc_stream_t stream);

/* The same NMS over n_segments independent box lists in one launch set (one list per image and pyramid level, say).  Segment s is
 * boxes[seg_offsets[s] .. + seg_counts[s]) (device int32 arrays; counts <= max_n <= 4096), ALREADY in descending score order.
 * keep[seg_offsets[s] + i] = index inside the segment of its i-th survivor (ascending = score order), n_keep[s] their count.
 * max_keep > 0: only the first max_keep survivors of every segment are wanted (boxlist_nms's max_proposals): the walk stops there. */
size_t vidc_nms_segmented_scratch_bytes(int n_segments, int max_n);
int vidc_nms_segmented(const float* boxes_xyxy, const int32_t* seg_offsets, const int32_t* seg_counts, int n_segments, int max_n,
                       float threshold, int inclusive, int max_keep, int32_t* keep, int32_t* n_keep, void* scratch, vidc_stream_t stream);

/* ROIAlign forward (csrc/cuda/ROIAlign_cuda.cu:65-176, csrc/cpu/ROIAlign_cpu.cpp:17-218; Caffe2 semantics: no half-pixel shift,
 * ROIs smaller than 1x1 forced to 1x1, sampling_ratio <= 0 -> ceil(roi / pooled) samples per bin and axis).
 * x: NHWC [N][H][W][ldx] (C channels used); rois float [K][5] = (batch index, x1, y1, x2, y2); y: [K][pooled_h][pooled_w][C]. */
int vidc_roi_align_forward(const float* x_nhwc, const float* rois, float* y, int K, int C, int H, int W, int ldx, int pooled_h,
                           int pooled_w, float spatial_scale, int sampling_ratio, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * plane-mask detector: everything of COCODemo.run_on_tensor that is not a convolution   (plane_mask_detection/demo/predictor.py:143-150;
 * SURVEY §8f-1).  Static shapes: R proposal / detection slots per image with the live counts in device memory; no host synchronisation.
 * ---------------------------------------------------------------------------------------------- */

/* demo/predictor.py:101-118,143-144 (uint8 cast of 255*image, BGR, x255, - PIXEL_MEAN) + to_image_list's zero padding to Hp x Wp, written
 * as the im2col matrix of the 7x7 / stride 2 / pad 3 stem: cols [B][Hp/2][Wp/2][160], column (kh*7+kw)*3 + c (c = B,G,R), 147..159 = 0. */
int vidc_det_stem_im2col(const float* image01_nchw, float* cols, int B, int H, int W, int Hp, int Wp, float mean_b, float mean_g,
                         float mean_r, vidc_stream_t stream);
/* F.interpolate(x, scale_factor=2, mode="nearest") of the FPN top-down path (modeling/backbone/fpn.py:66-72), NHWC. */
int vidc_upsample_nearest2x(const float* x, float* y, int B, int h, int w, int C, int ldx, int ldy, vidc_stream_t stream);

/* The `use_mask=True` branch of SurfaceNormalPrediction.forward (networks/surface_normal.py:150-162): y = x * mask, where mask =
 * (r + g + b > 1e-2) of the warped image (NCHW [B][3][H][W]) nearest-resized to the h x w feature map x (NHWC); the mask is not stored. */
int vidc_mask_scale(const float* x, const float* image_nchw, float* y, int B, int h, int w, int C, int ldx, int ldy, int H, int W,
                    vidc_stream_t stream);
/* RPNPostProcessor.forward_for_single_feature_map up to NMS (modeling/rpn/inference.py:74-100): rpn_map [B][h][w][ld] holds A objectness
 * logits then 4A box deltas per pixel; sigmoid, the k = min(pre_nms_top_n, A*h*w) best in descending order (ties: lower (h,w,a) index
 * first), anchors (cell_anchors_host [A][4], host memory, + the pixel's stride offset), BoxCoder(1,1,1,1).decode, clip to the image.
 * boxes [B][out_stride][4] / scores [B][out_stride]: the first k entries of every image are written. */
int vidc_rpn_topk_decode(const float* rpn_map, int B, int h, int w, int ld, int A, int stride, const float* cell_anchors_host,
                         int pre_nms_top_n, int img_h, int img_w, float* boxes, float* scores, long long out_stride, vidc_stream_t stream);
/* The same for all pyramid levels in ONE launch (grid = images x levels): maps_host [n_levels] device pointers, hw_host [n_levels][2],
 * strides_host [n_levels], cell_anchors_host [n_levels][A][4], level_offsets_host [n_levels] = first slot of the level inside a row of
 * boxes / scores (all host arrays).  Level l writes its k_l = min(pre_nms_top_n, A*h_l*w_l) entries at level_offsets_host[l]. */
int vidc_rpn_topk_decode_levels(const float* const* maps_host, const int32_t* hw_host, const int32_t* strides_host,
                                const float* cell_anchors_host, const int32_t* level_offsets_host, int n_levels, int B, int ld, int A,
                                int pre_nms_top_n, int img_h, int img_w, float* boxes, float* scores, long long out_stride,
                                vidc_stream_t stream);
/* The first per_level NMS survivors of every level (keep / n_keep as written by vidc_nms per level: keep [B][slots] at the level's
 * offset, indices relative to it; n_keep [B][n_levels]), then the `total` best over all levels (modeling/rpn/inference.py:103-108,
 * 148-190).  level_offsets_host [n_levels + 1] (host memory).  proposals [B][total][4] (zero-filled beyond n_proposals[b]). */
int vidc_rpn_select(const float* boxes, const float* scores, const int32_t* keep, const int32_t* n_keep, int B, int n_levels,
                    const int32_t* level_offsets_host, int per_level, int total, float* proposals, float* proposal_scores,
                    int32_t* n_proposals, vidc_stream_t stream);
/* Pooler.forward (modeling/poolers.py:11-122): LevelMapper + ROIAlign from the mapped level of P2..P5.  feats_host: 4 device pointers to
 * NHWC [B][h][w][C] maps (host array), hw_host [4][2]; boxes [B][R][4] (image index = slot / R); y [B*R][pooled][pooled][C]. */
int vidc_roi_align_fpn(const float* const* feats_host, const int32_t* hw_host, int n_levels, int C, const float* boxes, int B, int R,
                       int pooled, int sampling_ratio, float* y, vidc_stream_t stream);
/* PostProcessor.forward / filter_results for the two-class head (modeling/roi_heads/box_head/inference.py:47-146), in two steps around
 * vidc_nms.  head_out [B*R][ld]: 2 class logits, 8 box deltas.  Candidates = plane-class probability > score_thresh, decoded with
 * BoxCoder(10,10,5,5) and clipped, in descending score order (cand_src = proposal slot); unused slots hold far-away unit boxes. */
int vidc_det_candidates(const float* head_out, int ld, const float* proposals, const int32_t* n_proposals, int B, int R, int img_h, int img_w,
                        float score_thresh, float* cand_boxes, float* cand_scores, int32_t* cand_src, int32_t* n_cand, vidc_stream_t stream);
/* NMS survivors (keep / n_keep [B][R] / [B] from vidc_nms over the R candidate slots) in ascending proposal order, like the reference. */
int vidc_det_select(const float* cand_boxes, const float* cand_scores, const int32_t* cand_src, const int32_t* n_cand, const int32_t* keep,
                    const int32_t* n_keep, int B, int R, float* det_boxes, float* det_scores, int32_t* n_det, vidc_stream_t stream);
/* MaskPostProcessor + Masker(threshold, padding 1) (modeling/roi_heads/mask_head/inference.py:27-49, 86-150): sigmoid of the detection's
 * class channel `cls`, zero border, box expansion, bilinear resize (align_corners=False) to the integer box, threshold, paste.
 * mask_logits [B*R][M/2][M/2*4][ld]: the 2x2 transposed conv as a 1x1 conv to 4 sub-pixel channel blocks; pasted [B][R][H][W] uint8. */
int vidc_mask_paste(const float* mask_logits, int ld, int cls, int M, const float* det_boxes, const int32_t* n_det, int B, int R, int H, int W,
                    float thresh, uint8_t* pasted, vidc_stream_t stream);
/* select_top_predictions + overlay_mask (demo/predictor.py:201-220, 253-323): detections with score > confidence in descending score
 * order, the biggest 4-connected component of each mask, those of at least min_fraction * H * W pixels numbered 1.. by descending size
 * and painted in that order.  inst [B][H][W] uint8. */
size_t vidc_instance_map_scratch_bytes(int B, int R, int H, int W);
int vidc_instance_map(const uint8_t* pasted, const float* det_scores, const int32_t* n_det, int B, int R, int H, int W, float confidence,
                      float min_fraction, uint8_t* inst, void* scratch, vidc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * programs: a whole network (or the whole frame) as one native call / one hipGraph
 * ---------------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------------
 * Training step of the depth-completion network (network_run.py:158-191, 228-254): what
 * `_run_training_iteration` needs beyond the inference kernels.  Activations NHWC fp32 rows [M][ld] with C channels used.
 * `scratch`: vidc_train_scratch_bytes(M, C) bytes unless stated.  Per-channel reductions: fp64, fixed order.
 * ---------------------------------------------------------------------------------------------- */
size_t vidc_train_scratch_bytes(long long M, int C);

/* nn.BatchNorm2d in train() mode (+ the ReLU that follows it when relu != 0): batch mean / biased variance over the M rows,
 * y = (x - mean) * invstd * gamma + beta; running_mean/var (may be NULL) <- (1 - momentum) * old + momentum * (mean, unbiased var);
 * save_mean / save_rstd [C] are kept for the backward.  y_bf16 (may be NULL): additionally the output rounded to bf16 as dense rows of C
 * values -- the operand the next conv of a VIDC_PREC_BF16 step reads (saves its vidc_cast_bf16 launch). */
int vidc_bn_train_forward(const float* x, float* y, long long M, int C, int ldx, int ldy, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float eps, float momentum, int relu, float* save_mean, float* save_rstd,
                          void* y_bf16, void* scratch, vidc_stream_t stream);
/* The same with the Bottleneck tail in the apply pass: y = relu?(BatchNorm(x) + residual) (residual NHWC with channel stride ldr, or NULL);
 * the BatchNorm value is rounded to fp32 before the sum, so the result equals vidc_bn_train_forward followed by vidc_add_rows. */
int vidc_bn_train_forward_add(const float* x, float* y, long long M, int C, int ldx, int ldy, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float eps, float momentum, int relu, float* save_mean, float* save_rstd,
                              void* y_bf16, const float* residual, int ldr, void* scratch, vidc_stream_t stream);
/* vidc_bn_train_forward_add for an x that is the output of a conv launched with VIDC_STATS_OUT: `conv_stats` = that conv's partials
 * (ceil(M / 32) x 2 x C doubles); the partial-sum pass over x is skipped.  scratch: 2 * C doubles. */
int vidc_bn_train_forward_stats(const float* x, float* y, long long M, int C, int ldx, int ldy, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float eps, float momentum, int relu, float* save_mean, float* save_rstd,
                                void* y_bf16, const float* residual, int ldr, const void* conv_stats, void* scratch, vidc_stream_t stream);
/* Its backward.  y_relu: the forward output when a ReLU followed (its mask is applied to dy), else NULL.  dx may alias dy.
 * dx_bf16 (may be NULL): dx as dense bf16 rows as well (the dgrad conv of the layer in front reads it). */
int vidc_bn_train_backward(const float* dy, const float* x, const float* y_relu, float* dx, long long M, int C, int lddy, int ldx, int ldy,
                           int lddx, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                           void* dx_bf16, void* scratch, vidc_stream_t stream);
/* The same, which can also write dx TRANSPOSED as plain bf16 rows dx_bf16_t[C][Mp] (Mp = M rounded up to a multiple of 64, zeros past
 * M; may be NULL): when the BatchNorm follows a conv, dx is that conv's dY and this is the left operand of its weight-gradient GEMM --
 * what vidc_im2col_transposed(dY, KH = KW = 1, split = 2) would build with one more launch and one more pass over dY.  dx may be NULL when
 * both bf16 forms are written (a stride-1 conv without bias reads its dY only through them). */
int vidc_bn_train_backward_t(const float* dy, const float* x, const float* y_relu, float* dx, long long M, int C, int lddy, int ldx, int ldy,
                             int lddx, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                             void* dx_bf16, void* dx_bf16_t, int Mp, void* scratch, vidc_stream_t stream);
/* out[c] = sum over rows of dy[.][c]: the bias gradient of a convolution. */
int vidc_colsum(const float* dy, long long M, int C, int ld, float* out, void* scratch, vidc_stream_t stream);
/* y = a + b, ReLU optional (Bottleneck: relu(bn3(conv3(.)) + identity); decoder: z1 + z2 + z3 + z4). */
int vidc_add_rows(const float* a, const float* b, float* y, long long M, int C, int lda, int ldb, int ldy, int relu, vidc_stream_t stream);
/* The same; y_bf16 (may be NULL): additionally the result rounded to bf16 as dense rows of C values (the operand copy the next convs of a
 * VIDC_PREC_BF16 step read, as vidc_bn_train_forward's y_bf16). */
int vidc_add_rows_bf16(const float* a, const float* b, float* y, long long M, int C, int lda, int ldb, int ldy, int relu, void* y_bf16,
                       vidc_stream_t stream);
/* dx (accumulate ? += : =) dy * (y > 0); y == NULL: plain copy / accumulate. */
int vidc_relu_backward(const float* dy, const float* y, float* dx, long long M, int C, int lddy, int ldy, int lddx, int accumulate,
                       vidc_stream_t stream);
/* Backward of vidc_maxpool3x3s2 (first maximum of a window in scan order takes the gradient, like torch). x: the pool's input. */
int vidc_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int ldx, int lddy, int lddx,
                               vidc_stream_t stream);
/* Backward of vidc_upsample_bilinear_ac: dy [B][H][W] rows -> dx [B][h][w] rows (gather form, deterministic). */
int vidc_upsample_bilinear_ac_backward(const float* dy, float* dx, int B, int h, int w, int C, int lddy, int lddx, int H, int W,
                                       vidc_stream_t stream);
/* Backward of the padded 1x1 head conv (depth_completion.py:145: Conv2d(192, 1, 1, padding=1)): g_low [B][h+2][w+2] -> dx NHWC
 * [B][h][w][C], dw [C], dbias [1].  scratch: vidc_head_backward_scratch_bytes. */
size_t vidc_head_backward_scratch_bytes(int B, int h, int w, int C);
int vidc_head_backward(const float* g_low, const float* x, const float* wgt, float* dx, float* dw, float* dbias, int B, int h, int w, int C,
                       int ldx, int lddx, void* scratch, vidc_stream_t stream);
/* `_network_loss` (network_run.py:163-173): *loss = sum over gt > 0 of |pred - gt| / hw (hw = H*W of one image), dpred = its
 * gradient, terms = the per-pixel terms (n floats of scratch the caller owns); scratch: n / 512 + 8 doubles. */
int vidc_masked_l1_loss(const float* pred, const float* gt, long long n, int hw, double* loss, float* dpred, float* terms, void* scratch,
                        vidc_stream_t stream);
/* torch.optim.Adam.step (defaults: no weight decay, no amsgrad) over a flat buffer; step = 1, 2, ... */
int vidc_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, int step,
                   vidc_stream_t stream);
/* Process-wide switch of the training BatchNorms (default 0): with 1 the final reduction of the per-chunk channel sums runs inside the
 * prologue of the kernel that consumes it (apply / backward-apply: one launch less per BatchNorm pass on the small maps) instead of in a
 * launch of its own.  Both forms add the same numbers in the same order (bit-identical results); the folded form measured 6 % slower
 * per step on MI355X (cross-XCD reads of the chunk sums in every workgroup's prologue), so it is an opt-in kept for measurements.
 * enable < 0: query only.  Returns the previous value. */
int vidc_train_bn_fold(int enable);
/* A range of the flat gradient buffer rounded to bf16 (round to nearest even) for the cross-rank SUM, and widened back afterwards:
 * halves the bytes a ring all-reduce moves over xGMI (training.GradientBuckets, VIDC_TRAIN_GRAD_BF16=1; the reference sums its
 * replicas' gradients in fp32 inside DataParallel, network_run.py:97-99 -- this is an opt-in).  Both buffers 16-byte aligned. */
int vidc_grad_narrow_bf16(const float* x, void* y_bf16, long long n, vidc_stream_t stream);
int vidc_grad_widen_bf16(const void* x_bf16, float* y, long long n, vidc_stream_t stream);
/* dgrad: the gradient w.r.t. a conv's input is vidc_conv2d_bn_act of dY (stride 1, pad KH-1-pad) with these weights
 * ([Cin][Cout/32][KH][KW][32], kernel flipped); a stride-s conv first spreads dY over the input grid with vidc_zero_stuff. */
int vidc_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, vidc_stream_t stream);
/* HOST function (no device work): numpy.random.RandomState.permutation(n)[0:k] replayed on the generator's own MT19937 state (key624 =
 * get_state()[1], *pos = get_state()[2]), advancing it exactly as numpy does -- the n - 1 bounded draws of the Fisher-Yates shuffle.  The
 * reference draws its RANSAC hypotheses this way per plane (main.py:43, :78); this is its largest item of host time per frame.  idx_out: k
 * int32, scratch: n int32.  n < 2^31. */
int vidc_host_mt19937_permutation_prefix(uint32_t* key624, int32_t* pos, long long n, int k, int32_t* idx_out, int32_t* scratch);
/* All conv weights of a network re-packed by one launch (they move every optimizer step).  `items_device`: n_items descriptors in
 * device memory, sorted by block_begin; item i owns the vidc_pack_item_blocks(...) workgroups from block_begin_i on (0 = unsupported
 * shape: kernels up to 3x3, the K-side channel count a multiple of the unit); total_blocks = their sum.  kind: 0 forward fp32
 * (vidc_pack_conv_weight), 1 dgrad fp32 (vidc_pack_conv_weight_dgrad),
 * 2 / 3 the same two in split-bf16 (vidc_pack_conv_weight_bf16x3; dgrad rows split like vidc_split_bf16x3), 4 / 5 in plain bf16 for
 * VIDC_PREC_BF16 ([row][channels/64][KH][KW][64 x bf16]; the K-side channel count a multiple of 64). */
typedef struct vidc_pack_item {
    const float* w;          /* OIHW parameter                                   */
    float* packed;           /* Cout*Cin*KH*KW floats                            */
    int32_t Cout, Cin, KH, KW;
    int32_t kind, reserved;
    int64_t block_begin;
} vidc_pack_item;
long long vidc_pack_item_blocks(int Cout, int Cin, int KH, int KW, int kind);
int vidc_pack_conv_weights_batched(const vidc_pack_item* items_device, int n_items, long long total_blocks, vidc_stream_t stream);
int vidc_zero_stuff(const float* dy, float* z, int B, int Ho, int Wo, int C, int lddy, int stride, int H, int W, vidc_stream_t stream);
/* wgrad: dw_oihw[co][ci][kh][kw] = sum over output pixels of dy[m][co] * x[pixel(m) at the tap][ci], on v_mfma_f32_32x32x2_f32 with the
 * pixels as the reduction index.  scratch: vidc_conv_wgrad_scratch_bytes. */
size_t vidc_conv_wgrad_scratch_bytes(int B, int Ho, int Wo, int Cout, int Cin, int KH, int KW);
int vidc_conv_wgrad(const float* dy, const float* x, float* dw_oihw, int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int lddy,
                    int KH, int KW, int stride, int pad, void* scratch, vidc_stream_t stream);
/* wgrad as a GEMM on the conv kernel: dW[co][tap][ci] = sum_m dY^T[co][m] * Xt[tap*C + ci][m] is the 1x1 case of vidc_conv2d_bn_act with
 * "activations" = the rows of dY^T and "weights" = the rows of Xt (both K-contiguous, K = pixels).  vidc_im2col_transposed writes
 * xt[(tap*C + c)][m] = x[b, oy*s - p + kh, ox*s - p + kw, c] (0 outside the image and for m >= B*Ho*Wo; rows Mp long, Mp % 32 == 0);
 * with KH = KW = 1, stride 1, pad 0 it transposes dY; split = 1 writes the rows as split-bf16 operands (vidc_split_bf16x3's layout) for
 * a VIDC_PREC_BF16X3 GEMM, split = 2 as plain bf16 rows (Mp % 64 == 0) for a VIDC_PREC_BF16 one.  split + 4: the rows in CHANNEL-major
 * order, xt[(c*KH*KW + tap)][m] -- the GEMM's output row [co][c*KH*KW + tap] is then the OIHW layout of the weight gradient itself and
 * can be written straight into the parameter's .grad.  vidc_wgrad_permute (tap-major order): dw_oihw[co][ci][tap] = tmp[co][tap*Cin + ci]. */
int vidc_im2col_transposed(const float* x, float* xt, int B, int H, int W, int C, int ldx, int Ho, int Wo, int KH, int KW, int stride, int pad,
                           int Mp, int split, vidc_stream_t stream);
/* The right operand of a 1x1 / stride-1 conv's weight-gradient GEMM in the plain-bf16 mode from the bf16 operand copy the forward read:
 * dense bf16 rows x[M][C] -> xt[C][Mp] (zeros for m >= M, Mp % 64 == 0, C % 8 == 0).  Same bits as vidc_im2col_transposed(split = 2) of
 * the fp32 tensor the copy was rounded from. */
int vidc_transpose_bf16(const void* x_bf16, void* xt_bf16, long long M, int C, int Mp, vidc_stream_t stream);
/* vidc_im2col_transposed(..., split = 2 + 4) from the bf16 operand copy of x (dense bf16 rows [B*H*W][C]) instead of the fp32 tensor: the
 * right operand of any conv's weight-gradient GEMM in the plain-bf16 mode at half the input bytes, same bits. */
int vidc_im2col_transposed_bf16(const void* x_bf16, void* xt_bf16, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                                int Mp, vidc_stream_t stream);
int vidc_wgrad_permute(const float* tmp, float* dw_oihw, int Cout, int Cin, int taps, vidc_stream_t stream);
/* wgrad of the 3x3 / stride-2 stem conv on the NCHW network input (Cin = 1 or 3). */
size_t vidc_stem_wgrad_scratch_bytes(int B, int Cin, int H, int W, int Cout);
int vidc_stem_wgrad(const float* dy, const float* x_nchw, float* dw_oihw, int B, int Cin, int H, int W, int Cout, int lddy, void* scratch,
                    vidc_stream_t stream);

enum vidc_op_kind { VIDC_OP_CONV = 1, VIDC_OP_STEM = 2, VIDC_OP_MAXPOOL = 3, VIDC_OP_UPSAMPLE = 4, VIDC_OP_HEAD = 5,
                    VIDC_OP_WARP_PARAMS = 6, VIDC_OP_WARP_FWD = 7, VIDC_OP_WARP_INV = 8, VIDC_OP_COPY = 9, VIDC_OP_SPLIT = 10,
                    VIDC_OP_AVGPOOL = 11, VIDC_OP_NORMALIZE = 12, VIDC_OP_DET_IM2COL = 13, VIDC_OP_NEAREST2X = 14,
                    /* 15: retired (persistent conv chain, removed in round 5) */ VIDC_OP_MASK = 16 /* vidc_mask_scale: p = x, image, y; i = B,h,w,C,ldx,ldy,H,W */,
                    VIDC_OP_WINO_IN = 17  /* vidc_winograd_input_transform: p = x, v; i = B,H,W,C,ldx,Cin,m,split,ldv */,
                    VIDC_OP_WINO_OUT = 18 /* vidc_winograd_output_transform: p = mm, y, y_split, scale1, shift1, scale2, shift2; i = B,Ho,Wo,C,Cout,ldy,m,flags,ldm */ };

typedef struct vidc_generic_args {   /* arguments of the non-conv launchers, in declaration order */
    const void* p[8];
    int32_t i[16];
    float f[8];
} vidc_generic_args;

typedef struct vidc_op {
    int32_t kind;      /* vidc_op_kind */
    int32_t stream_id; /* 0..VIDC_MAX_STREAMS-1: ops on different ids may overlap (fork/join by events) */
    int32_t wait_mask; /* bit s set: wait for everything previously issued on stream id s */
    int32_t reserved;
    union {
        vidc_conv_desc conv;
        vidc_generic_args g;
    } u;
} vidc_op;

#define VIDC_MAX_STREAMS 4
#define VIDC_MAX_SEGMENTS 4

typedef struct vidc_program vidc_program;
int vidc_program_create(const vidc_op* ops, int n_ops, vidc_program** out);
int vidc_program_run(vidc_program* p, vidc_stream_t stream);          /* eager: one launch per op            */
int vidc_program_capture(vidc_program* p, vidc_stream_t stream);      /* records a hipGraph of the program    */
int vidc_program_launch(vidc_program* p, vidc_stream_t stream);       /* replays the captured graph (all segments) */
/* Segments: ops [begin, end) of one planned program as their own eager run / hipGraph (segment 0..VIDC_MAX_SEGMENTS-1), so the
 * host can place other launches between parts of a program -- the reference does the same thing implicitly: its plane block
 * and enrichment loop (main.py:277-294) sit between the two networks of one _call_cnn. */
int vidc_program_run_range(vidc_program* p, vidc_stream_t stream, int begin, int end);
int vidc_program_capture_range(vidc_program* p, vidc_stream_t stream, int begin, int end, int segment);
int vidc_program_launch_segment(vidc_program* p, vidc_stream_t stream, int segment);
/* Times `iters` back-to-back executions with hipEvents on `stream`; ms_out[0] = average ms per execution,
 * and (if per_op_ms != NULL, eager mode) per-op average durations. */
int vidc_program_time(vidc_program* p, vidc_stream_t stream, int iters, int use_graph, float* ms_out, float* per_op_ms);
int vidc_program_destroy(vidc_program* p);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* VIDC_H */
