"""ctypes binding of libvidc.so (the C ABI declared in include/vidc.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
`build()` compiles it in-tree with hipcc for gfx950 (also used by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("VIDC_LIB_NAME", "libvidc.so"))      # (VIDC_LIB_NAME: A/B builds of tools/, never set in production)
CSRC = os.path.join(_HERE, "csrc")

WARP_PARAMS = 32
MAX_HYP = 300
PLANE_RECORD = 16
MAX_STREAMS = 4
MAX_SEGMENTS = 4
CLOCK_STAMP_WGS = 8      # VIDC_CLOCK_STAMP_WGS

# vidc_conv_flags / vidc_up_flags / vidc_op_kind / vidc_conv_tile
RELU1, AFFINE2, RELU2, RESIDUAL, RELU3, ACCUM, SPLIT_OUT, NO_F32_OUT, STATS_OUT, X_PLANAR_GROUPS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
UP_RELU, UP_ACCUM, UP_NO_F32_OUT = 1, 2, 4
OP_CONV, OP_STEM, OP_MAXPOOL, OP_UPSAMPLE, OP_HEAD, OP_WARP_PARAMS, OP_WARP_FWD, OP_WARP_INV, OP_COPY, OP_SPLIT, OP_AVGPOOL, OP_NORMALIZE, OP_DET_IM2COL, OP_NEAREST2X, _OP_RETIRED_15, OP_MASK, OP_WINO_IN, OP_WINO_OUT = range(1, 19)
TILE_AUTO = 0
TILE_NAMES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x128", 4: "64x64", 5: "64x64k2", 6: "32x64k2", 7: "32x32k4", 8: "32x128",
              9: "32x32k8", 10: "32x64k2d5", 11: "32x32k4d4", 12: "32x128d6", 13: "64x64k2d4", 14: "32x64k2L", 15: "32x64k2d5L", 16: "32x32k4d4L", 17: "64x64L", 18: "64x64k2d4L",
              19: "64x128L", 20: "128x64L", 21: "64x32k2", 22: "64x32k2d5", 23: "64x32k2d5L", 24: "128x128d3", 25: "128x128d3L", 26: "256x128", 27: "128x256",
              28: "32x64k2d2", 29: "64x64d2", 30: "32x32k4d2", 31: "64x128d2", 32: "64x32k2d2",
              33: "128x128d4P", 34: "128x128d3P", 35: "64x64d4P", 36: "128x64d4P", 37: "64x64k2d4P", 38: "64x32k2d5P", 39: "32x64k2d5P", 40: "g96x32s", 41: "g96x64s3", 42: "wino4f"}
TILE_COUNT = 43
TILE_WINO4_FUSED = 42
PREC_FP32, PREC_BF16X3, PREC_BF16 = 0, 1, 2
SPLITK_COUNTERS = 16384            # VIDC_SPLITK_COUNTERS: ticket counters at the head of a split-K workspace

_f32p = C.POINTER(C.c_float)


class PackItem(C.Structure):
    """vidc_pack_item (include/vidc.h)."""
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("kind", C.c_int32), ("reserved", C.c_int32), ("block_begin", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
        ("scale1", C.c_void_p), ("shift1", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p),
        ("residual", C.c_void_p), ("workspace", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("ldx", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("ldy", C.c_int32), ("ldr", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("flags", C.c_int32), ("groups", C.c_int32),
        ("x_gs", C.c_int64), ("w_gs", C.c_int64), ("y_gs", C.c_int64), ("r_gs", C.c_int64), ("p_gs", C.c_int64),
        ("tile", C.c_int32), ("splitk", C.c_int32), ("precision", C.c_int32), ("dilation", C.c_int32),
        ("y_split", C.c_void_p),
    ]


class GenericArgs(C.Structure):
    _fields_ = [("p", C.c_void_p * 8), ("i", C.c_int32 * 16), ("f", C.c_float * 8)]


class _OpUnion(C.Union):
    _fields_ = [("conv", ConvDesc), ("g", GenericArgs)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("stream_id", C.c_int32), ("wait_mask", C.c_int32), ("reserved", C.c_int32),
                ("u", _OpUnion)]


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
# name -> (restype, argtypes); every symbol include/vidc.h declares
SIGNATURES = {
    "vidc_version": (C.c_int, []),
    "vidc_last_error": (C.c_char_p, []),
    "vidc_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, _i]),
    "vidc_warp2dof_params": (C.c_int, [_vp, _vp, _i, _f, _f, _f, _f, _vp, _i, _i, _vp, _vp]),
    "vidc_warp2dof_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "vidc_warp2dof_inv_rot_norm": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp]),
    "vidc_pack_conv_weight": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vidc_split_bf16x3": (C.c_int, [_vp, _vp, C.c_longlong, _i, _i, _vp]),
    "vidc_pack_conv_weight_bf16x3": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vidc_conv2d_bn_act": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "vidc_conv2d_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "vidc_conv2d_plan": (C.c_int, [C.POINTER(ConvDesc)]),
    "vidc_train_scratch_bytes": (C.c_size_t, [C.c_longlong, _i]),
    "vidc_bn_train_forward": (C.c_int, [_vp, _vp, C.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "vidc_bn_train_forward_add": (C.c_int, [_vp, _vp, C.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "vidc_bn_train_forward_stats": (C.c_int, [_vp, _vp, C.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "vidc_bn_train_backward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vidc_bn_train_backward_t": (C.c_int, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "vidc_colsum": (C.c_int, [_vp, C.c_longlong, _i, _i, _vp, _vp, _vp]),
    "vidc_add_rows": (C.c_int, [_vp, _vp, _vp, C.c_longlong, _i, _i, _i, _i, _i, _vp]),
    "vidc_add_rows_bf16": (C.c_int, [_vp, _vp, _vp, C.c_longlong, _i, _i, _i, _i, _i, _vp, _vp]),
    "vidc_relu_backward": (C.c_int, [_vp, _vp, _vp, C.c_longlong, _i, _i, _i, _i, _i, _vp]),
    "vidc_maxpool3x3s2_backward": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_upsample_bilinear_ac_backward": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_head_backward_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "vidc_head_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vidc_masked_l1_loss": (C.c_int, [_vp, _vp, C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp]),
    "vidc_clock_stamp": (C.c_int, [_vp, _vp]),
    "vidc_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_longlong, _f, _f, _f, _f, _i, _vp]),
    "vidc_train_bn_fold": (C.c_int, [_i]),
    "vidc_grad_narrow_bf16": (C.c_int, [_vp, _vp, C.c_longlong, _vp]),
    "vidc_grad_widen_bf16": (C.c_int, [_vp, _vp, C.c_longlong, _vp]),
    "vidc_pack_conv_weight_dgrad": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vidc_cast_bf16": (C.c_int, [_vp, _vp, C.c_longlong, _i, _i, _vp]),
    "vidc_host_mt19937_permutation_prefix": (C.c_int, [_vp, _vp, C.c_longlong, _i, _vp, _vp]),
    "vidc_pack_item_blocks": (C.c_longlong, [_i, _i, _i, _i, _i]),
    "vidc_pack_conv_weights_batched": (C.c_int, [_vp, _i, C.c_longlong, _vp]),
    "vidc_zero_stuff": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_conv_wgrad_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i, _i, _i, _i]),
    "vidc_conv_wgrad": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vidc_im2col_transposed": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_wgrad_permute": (C.c_int, [_vp, _vp, _i, _i, _i, _vp]),
    "vidc_transpose_bf16": (C.c_int, [_vp, _vp, C.c_longlong, _i, _i, _vp]),
    "vidc_im2col_transposed_bf16": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_stem_wgrad_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "vidc_stem_wgrad": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vidc_winograd_tiles": (C.c_int, [_i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vidc_winograd_weight_transform": (C.c_int, [_vp, _vp, _i, _i, _i, _vp]),
    "vidc_winograd_weight_pack_fused": (C.c_int, [_vp, _vp, _i, _i, _vp]),
    "vidc_winograd_input_transform": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_winograd_output_transform": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_stem_conv3x3s2": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "vidc_stem_conv3x3s2_warped": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _f, _f, _i, _vp]),
    "vidc_maxpool3x3s2": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vidc_upsample_bilinear_ac": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vidc_avgpool2d": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_normalize_nchw": (C.c_int, [_vp, _vp, _i, _i, _i, _vp]),
    "vidc_head_conv1x1_upsample": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_plane_scratch_bytes": (C.c_size_t, [_i, _i, _i]),
    "vidc_plane_ransac_normal": (C.c_int, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "vidc_plane_offset": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vidc_plane_offset_dense": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vidc_plane_project_depth": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "vidc_plane_info_count": (C.c_int, [_i, _i]),
    "vidc_plane_finalize": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "vidc_plane_block": (C.c_int, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vidc_enrich_scatter": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "vidc_enrich_scatter_from": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "vidc_resize_coeffs": (C.c_int, [_i, _i, _vp, _vp, _i, C.POINTER(C.c_int)]),
    "vidc_resize_bilinear_u8_to_chw": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "vidc_rasterize_sparse_depth": (C.c_int, [_vp, _vp, _i, C.c_double, C.c_double, C.c_double, C.c_double, _vp, _i, _i, _vp]),
    "vidc_nearest_table": (C.c_int, [_i, _i, _vp]),
    "vidc_resize_nearest_u16_depth": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, C.c_float, _vp]),
    "vidc_depth_metrics_scratch_bytes": (C.c_size_t, [C.c_longlong]),
    "vidc_depth_metrics": (C.c_int, [_vp, _vp, C.c_longlong, _vp, _i, _vp, _vp]),
    "vidc_depth_to_mm_u32": (C.c_int, [_vp, _vp, C.c_longlong, _vp]),
    "vidc_normal_metrics": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "vidc_hist_u16": (C.c_int, [_vp, C.c_longlong, _i, _vp, _vp]),
    "vidc_nms_scratch_bytes": (C.c_size_t, [_i]),
    "vidc_nms": (C.c_int, [_vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "vidc_nms_segmented_scratch_bytes": (C.c_size_t, [_i, _i]),
    "vidc_nms_segmented": (C.c_int, [_vp, _vp, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "vidc_roi_align_forward": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "vidc_det_stem_im2col": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "vidc_upsample_nearest2x": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_mask_scale": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vidc_rpn_topk_decode": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, C.c_longlong, _vp]),
    "vidc_rpn_topk_decode_levels": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, C.c_longlong, _vp]),
    "vidc_rpn_select": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "vidc_roi_align_fpn": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vidc_det_candidates": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "vidc_det_select": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "vidc_mask_paste": (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "vidc_instance_map_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "vidc_instance_map": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "vidc_program_create": (C.c_int, [C.POINTER(Op), _i, C.POINTER(_vp)]),
    "vidc_program_run": (C.c_int, [_vp, _vp]),
    "vidc_program_capture": (C.c_int, [_vp, _vp]),
    "vidc_program_launch": (C.c_int, [_vp, _vp]),
    "vidc_program_run_range": (C.c_int, [_vp, _vp, _i, _i]),
    "vidc_program_capture_range": (C.c_int, [_vp, _vp, _i, _i, _i]),
    "vidc_program_launch_segment": (C.c_int, [_vp, _vp, _i]),
    "vidc_program_time": (C.c_int, [_vp, _vp, _i, _i, _f32p, _f32p]),
    "vidc_program_destroy": (C.c_int, [_vp]),
}

_lib = None


def build(force=False, verbose=False):
    """Compile libvidc.so in-tree: hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".map")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "vidc.h"))
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cmd = ["make", "-C", CSRC, "-j%d" % min(8, os.cpu_count() or 1)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libvidc.so failed (hipcc, gfx950)")
    return LIB_PATH


def lib():
    """The loaded library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvidc.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`. "
                               "There is no CPU/eager fallback for the HIP path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)     # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if L.vidc_version() != 1:
            raise RuntimeError("libvidc.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libvidc %s failed (status %d): %s" % (what, rc, lib().vidc_last_error().decode()))


def ptr(t):
    """Raw device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def current_stream():
    """Raw hipStream_t of torch's current stream on the current device.  Called once per launch from Python: the private C accessors
    (no Stream object, no device-index resolution: ~0.3 us instead of ~7 us, 15 calls per frame in the stream modes) when this torch has
    them, torch.cuda.current_stream() otherwise."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get_raw, get_dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get_raw is not None and get_dev is not None:
            _raw_stream = lambda: get_raw(get_dev())          # noqa: E731
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
    return _raw_stream()
