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
    torch.ops.vidc.stem_conv3x3s2(x_nchw, w_oihw, relu)                       surface_normal.py:36 (conv1_1, no BN)
    torch.ops.vidc.maxpool3x3s2(x_nhwc)                                       torchvision ResNet.maxpool
    torch.ops.vidc.upsample_bilinear_ac(x_nhwc, Ho, Wo, relu)                 nn.UpsamplingBilinear2d (align_corners=True)
    torch.ops.vidc.head_conv1x1_upsample(x_nhwc, w, bias, pad, Ho, Wo, relu)  depth_completion.py:141-147 / surface_normal.py:140-145
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


OPS = ("warp2dof_fwd", "warp2dof_inv_rot_norm", "conv2d_bn_act", "stem_conv3x3s2", "maxpool3x3s2", "upsample_bilinear_ac",
       "head_conv1x1_upsample")
