"""`torch.ops.vidc.*`: the C entry points of libvidc.so registered as PyTorch custom operators (north_star: "exposed as
custom torch ops through a thin C-ABI extension").

Every operator is registered for the GPU dispatch key only (`device_types="cuda"`, which is HIP on ROCm): calling one with
CPU tensors raises PyTorch's "no kernel for backend CPU" error -- there is deliberately no CPU implementation.  Shape
functions (`register_fake`) are provided so the operators can be traced/exported; they allocate nothing on a device.

The networks themselves do not dispatch through these (a network is ONE `vidc_program_run`/hipGraph replay, engine.py);
the operators are the per-op surface for callers that compose the kernels with other PyTorch code:

    torch.ops.vidc.warp2dof_fwd(x, g, a, fx, fy, cx, cy, align_corners)      warping_2dof_alignment.py:108-156
    torch.ops.vidc.warp2dof_inv_rot_norm(x, g, a, fx, fy, cx, cy, align_corners, normalize)   :216-255 (+ surface_normal.py:170)
    torch.ops.vidc.conv2d_bn_act(x_nhwc, w_oihw, scale, shift, stride, pad, relu, precision)  Conv2d+BatchNorm2d(eval)+ReLU
    torch.ops.vidc.conv3x3_winograd(x_nhwc, w_oihw, scale, shift, m, relu, precision)         the 3x3 / stride-1 layers as Winograd F(m x m, 3x3)
    torch.ops.vidc.stem_conv3x3s2(x_nchw, w_oihw, relu)                       surface_normal.py:36 (conv1_1, no BN)
    torch.ops.vidc.maxpool3x3s2(x_nhwc)                                       torchvision ResNet.maxpool
    torch.ops.vidc.upsample_bilinear_ac(x_nhwc, Ho, Wo, relu)                 nn.UpsamplingBilinear2d (align_corners=True)
    torch.ops.vidc.head_conv1x1_upsample(x_nhwc, w, bias, pad, Ho, Wo, relu)  depth_completion.py:141-147 / surface_normal.py:140-145
    torch.ops.vidc.plane_ransac_normal(normals, ids, slots, hyp_pix)          main.py:38-62 (+ the write-back of :157)
    torch.ops.vidc.plane_offset(homo, depth, slots, inlier_mask, counts, scratch)       main.py:68-101, 162-173
    torch.ops.vidc.plane_project_depth(homo, slots, inlier_mask, scratch, records, plane_depth)   main.py:110-127
    torch.ops.vidc.plane_finalize(depth, plane_depth, records)                main.py:186-187 + the candidate counts of :287-289
    torch.ops.vidc.enrich_scatter(plane_depth, sparse_depth, sub, sub_offsets, chunk_base)        main.py:286, 290-294
(the host-side draws that feed them -- np.random.permutation / randint in the reference's order -- are plane.draw_normal_hypotheses
and plane.draw_enrichment; plane.PlaneBlock is the composition the pipeline uses)
"""
from typing import Tuple

import torch

from . import ops as _ops
from .networks.warping_2dof_alignment import Warping2DOFAlignment

_DEV = "cuda"


def _warper(x, fx, fy, cx, cy, align_corners):
    return Warping2DOFAlignment(fx, fy, cx, cy, align_corners=align_corners, device=x.device)


@torch.library.custom_op("vidc::warp2dof_fwd", mutates_args=(), device_types=_DEV)
def warp2dof_fwd(x: torch.Tensor, gravity: torch.Tensor, aligned: torch.Tensor, fx: float, fy: float, cx: float, cy: float,
                 align_corners: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Cg_H_C (B,3,3), warped x (B,C,H,W)) -- Warping2DOFAlignment.warp_with_gravity_center_aligned."""
    h, y = _warper(x, fx, fy, cx, cy, align_corners).warp_with_gravity_center_aligned(x, gravity, aligned)
    return h, y


@warp2dof_fwd.register_fake
def _(x, gravity, aligned, fx, fy, cx, cy, align_corners):
    return x.new_empty((x.shape[0], 3, 3)), torch.empty_like(x)


@torch.library.custom_op("vidc::warp2dof_inv_rot_norm", mutates_args=(), device_types=_DEV)
def warp2dof_inv_rot_norm(x: torch.Tensor, gravity: torch.Tensor, aligned: torch.Tensor, fx: float, fy: float, cx: float, cy: float,
                          align_corners: bool, normalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Cg_H_C, R^T-rotated inverse-warped normals [unit length if normalize]) -- inverse_warp_normal_image_with_gravity_center_aligned."""
    h, z = _warper(x, fx, fy, cx, cy, align_corners).inverse_warp_normal_image_with_gravity_center_aligned(x, gravity, aligned,
                                                                                                            normalize=normalize)
    return h, z


@warp2dof_inv_rot_norm.register_fake
def _(x, gravity, aligned, fx, fy, cx, cy, align_corners, normalize):
    return x.new_empty((x.shape[0], 3, 3)), torch.empty_like(x)


@torch.library.custom_op("vidc::conv2d_bn_act", mutates_args=(), device_types=_DEV)
def conv2d_bn_act(x_nhwc: torch.Tensor, w_oihw: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, stride: int, pad: int,
                  relu: bool, precision: int) -> torch.Tensor:
    """relu?(conv(x, w) * scale + shift) on NHWC activations; scale/shift = the folded bias + eval-mode BatchNorm
    (engine.fold_bn); precision 0 = exact fp32 MFMA, 1 = bf16x3 (vidc_conv_precision)."""
    pack = _ops.pack_conv_weight_bf16x3 if precision == 1 else _ops.pack_conv_weight
    return _ops.conv2d_bn_act(x_nhwc, pack(w_oihw), scale, shift, w_oihw.shape[2], w_oihw.shape[3], stride=stride, pad=pad,
                              relu1=relu, precision=precision)


@conv2d_bn_act.register_fake
def _(x_nhwc, w_oihw, scale, shift, stride, pad, relu, precision):
    B, H, W, _c = x_nhwc.shape
    co, _ci, kh, kw = w_oihw.shape
    return x_nhwc.new_empty((B, (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1, co))


@torch.library.custom_op("vidc::conv3x3_winograd", mutates_args=(), device_types=_DEV)
def conv3x3_winograd(x_nhwc: torch.Tensor, w_oihw: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, m: int, relu: bool,
                     precision: int) -> torch.Tensor:
    """relu?(conv3x3(x, w, stride 1, pad 1) * scale + shift) as Winograd F(m x m, 3x3), m = 2 or 4: input transform, the a*a
    transform-domain GEMMs as one grouped launch of the MFMA conv kernel, output transform with the epilogue (csrc/winograd.hip);
    the form the networks' 3x3 layers run in where the measured table says so.  precision as in conv2d_bn_act."""
    return _ops.conv3x3_winograd(x_nhwc, [w_oihw], scale.reshape(1, -1), shift.reshape(1, -1), m, relu1=relu, precision=precision)


@conv3x3_winograd.register_fake
def _(x_nhwc, w_oihw, scale, shift, m, relu, precision):
    B, H, W, _c = x_nhwc.shape
    return x_nhwc.new_empty((B, H, W, w_oihw.shape[0]))


@torch.library.custom_op("vidc::stem_conv3x3s2", mutates_args=(), device_types=_DEV)
def stem_conv3x3s2(x_nchw: torch.Tensor, w_oihw: torch.Tensor, relu: bool) -> torch.Tensor:
    return _ops.stem_conv3x3s2(x_nchw, w_oihw, relu=relu)


@stem_conv3x3s2.register_fake
def _(x_nchw, w_oihw, relu):
    B, _c, H, W = x_nchw.shape
    return x_nchw.new_empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, w_oihw.shape[0]))


@torch.library.custom_op("vidc::maxpool3x3s2", mutates_args=(), device_types=_DEV)
def maxpool3x3s2(x_nhwc: torch.Tensor) -> torch.Tensor:
    return _ops.maxpool3x3s2(x_nhwc)


@maxpool3x3s2.register_fake
def _(x_nhwc):
    B, H, W, Cc = x_nhwc.shape
    return x_nhwc.new_empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc))


@torch.library.custom_op("vidc::upsample_bilinear_ac", mutates_args=(), device_types=_DEV)
def upsample_bilinear_ac(x_nhwc: torch.Tensor, out_h: int, out_w: int, relu: bool) -> torch.Tensor:
    return _ops.upsample_bilinear_ac(x_nhwc, (out_h, out_w), relu=relu)


@upsample_bilinear_ac.register_fake
def _(x_nhwc, out_h, out_w, relu):
    return x_nhwc.new_empty((x_nhwc.shape[0], out_h, out_w, x_nhwc.shape[3]))


@torch.library.custom_op("vidc::head_conv1x1_upsample", mutates_args=(), device_types=_DEV)
def head_conv1x1_upsample(x_nhwc: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, pad: int, out_h: int, out_w: int,
                          relu: bool) -> torch.Tensor:
    """NCHW (B,Cout,out_h,out_w): 1x1 conv (+bias, zero padding `pad`) -> bilinear(align_corners=True) -> ReLU?"""
    return _ops.head_conv1x1_upsample(x_nhwc, w, bias, pad, (out_h, out_w), relu)[0]


@head_conv1x1_upsample.register_fake
def _(x_nhwc, w, bias, pad, out_h, out_w, relu):
    return x_nhwc.new_empty((x_nhwc.shape[0], w.shape[0], out_h, out_w))


# ---- plane block (csrc/plane.hip) ------------------------------------------------------------------------------------------------
from . import _lib as _L  # noqa: E402


def _st():
    return _L.current_stream()


@torch.library.custom_op("vidc::plane_ransac_normal", mutates_args=(), device_types=_DEV)
def plane_ransac_normal(normals: torch.Tensor, ids: torch.Tensor, slots: torch.Tensor, hyp_pix: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """normals (B,3,H,W) fp32 unit normals, ids (B,H,W) uint8 plane ids, slots (n,4) int32 = (image, id, hypothesis offset, count),
    hyp_pix int32 flat pixel indices of the hypotheses -> (inlier_mask (n,H*W) uint8, counts (n,300) int32, scratch uint8: the partial
    sums the next two stages read)."""
    B, _c, H, W = normals.shape
    n, HW = slots.shape[0], H * W
    lib = _L.lib()
    mask = torch.empty((n, HW), dtype=torch.uint8, device=normals.device)
    counts = torch.empty((n, _L.MAX_HYP), dtype=torch.int32, device=normals.device)
    scratch = torch.empty(lib.vidc_plane_scratch_bytes(n, B, HW), dtype=torch.uint8, device=normals.device)
    _L.check(lib.vidc_plane_ransac_normal(_L.ptr(normals.contiguous()), _L.ptr(ids.contiguous()), _L.ptr(slots.contiguous()), n,
                                          _L.ptr(hyp_pix.contiguous()), HW, _L.ptr(mask), _L.ptr(counts), _L.ptr(scratch), _st()), "plane_ransac_normal")
    return mask, counts, scratch


@plane_ransac_normal.register_fake
def _(normals, ids, slots, hyp_pix):
    B, _c, H, W = normals.shape
    n = slots.shape[0]
    nc = (H * W + 255) // 256
    return (normals.new_empty((n, H * W), dtype=torch.uint8), normals.new_empty((n, 300), dtype=torch.int32),
            normals.new_empty((n * nc * 9 * 4 + B * (H * W + 1) * 4 + 256,), dtype=torch.uint8))


@torch.library.custom_op("vidc::plane_offset", mutates_args=(), device_types=_DEV)
def plane_offset(homo: torch.Tensor, depth: torch.Tensor, slots: torch.Tensor, inlier_mask: torch.Tensor, counts: torch.Tensor,
                 scratch: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """homo (B,H,W,3), depth (B,1,H,W) sparse depth -> (records (n,16) fp32 [include/vidc.h], scratch passed on).  Planes with more than
    300 sparse points come back flagged (record[10] = -1): plane.PlaneBlock resolves them with the reference's host permutation."""
    B = homo.shape[0]
    n, HW = inlier_mask.shape
    sc = scratch.clone()
    rec = torch.empty((n, _L.PLANE_RECORD), dtype=torch.float32, device=homo.device)
    _L.check(_L.lib().vidc_plane_offset(_L.ptr(homo.contiguous()), _L.ptr(depth.contiguous()), _L.ptr(slots.contiguous()), n, B,
                                        _L.ptr(inlier_mask), _L.ptr(counts), HW, _L.ptr(sc), _L.ptr(rec), _st()), "plane_offset")
    return rec, sc


@plane_offset.register_fake
def _(homo, depth, slots, inlier_mask, counts, scratch):
    return homo.new_empty((inlier_mask.shape[0], 16)), torch.empty_like(scratch)


@torch.library.custom_op("vidc::plane_project_depth", mutates_args=(), device_types=_DEV)
def plane_project_depth(homo: torch.Tensor, slots: torch.Tensor, inlier_mask: torch.Tensor, scratch: torch.Tensor, records: torch.Tensor,
                        plane_depth: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """generate_depth_from_plane for every slot: (plane_depth with the valid planes written in, records with [10], [11] filled)."""
    n, HW = inlier_mask.shape
    out, rec, sc = plane_depth.contiguous().clone(), records.clone(), scratch.clone()
    _L.check(_L.lib().vidc_plane_project_depth(_L.ptr(homo.contiguous()), _L.ptr(slots.contiguous()), n, _L.ptr(inlier_mask), HW, _L.ptr(sc),
                                               _L.ptr(rec), _L.ptr(out), _st()), "plane_project_depth")
    return out, rec


@plane_project_depth.register_fake
def _(homo, slots, inlier_mask, scratch, records, plane_depth):
    return torch.empty_like(plane_depth), torch.empty_like(records)


@torch.library.custom_op("vidc::plane_finalize", mutates_args=(), device_types=_DEV)
def plane_finalize(depth: torch.Tensor, plane_depth: torch.Tensor, records: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(plane_depth with the original sparse depths restored (main.py:186-187), info int32: per 256-pixel chunk the number of
    plane_depth > 0 pixels of every image, then the number of flagged slots)."""
    B = depth.shape[0]
    HW = depth.numel() // B
    out = plane_depth.contiguous().clone()
    info = torch.empty(_L.lib().vidc_plane_info_count(B, HW), dtype=torch.int32, device=depth.device)
    _L.check(_L.lib().vidc_plane_finalize(_L.ptr(depth.contiguous()), _L.ptr(out), B, HW, _L.ptr(records.contiguous()), records.shape[0],
                                          _L.ptr(info), _st()), "plane_finalize")
    return out, info


@plane_finalize.register_fake
def _(depth, plane_depth, records):
    B = depth.shape[0]
    HW = depth.numel() // B
    return torch.empty_like(plane_depth), depth.new_empty((B * ((HW + 255) // 256) + 1,), dtype=torch.int32)


@torch.library.custom_op("vidc::enrich_scatter", mutates_args=(), device_types=_DEV)
def enrich_scatter(plane_depth: torch.Tensor, sparse_depth: torch.Tensor, sub: torch.Tensor, sub_offsets: torch.Tensor,
                   chunk_base: torch.Tensor) -> torch.Tensor:
    """clone(sparse_depth) with the sub[]-th nonzeros (row-major) of plane_depth copied in, per image (main.py:286-294)."""
    B = sparse_depth.shape[0]
    HW = sparse_depth.numel() // B
    out = torch.empty_like(sparse_depth, memory_format=torch.contiguous_format)
    _L.check(_L.lib().vidc_enrich_scatter_from(_L.ptr(plane_depth.contiguous()), _L.ptr(sparse_depth.contiguous()), _L.ptr(sub.contiguous()),
                                               _L.ptr(sub_offsets.contiguous()), _L.ptr(chunk_base.contiguous()), B, HW, _L.ptr(out), _st()),
             "enrich_scatter")
    return out


@enrich_scatter.register_fake
def _(plane_depth, sparse_depth, sub, sub_offsets, chunk_base):
    return torch.empty_like(sparse_depth)


OPS = ("plane_ransac_normal", "plane_offset", "plane_project_depth", "plane_finalize", "enrich_scatter", "warp2dof_fwd", "warp2dof_inv_rot_norm", "conv2d_bn_act", "stem_conv3x3s2", "maxpool3x3s2", "upsample_bilinear_ac",
       "head_conv1x1_upsample")
