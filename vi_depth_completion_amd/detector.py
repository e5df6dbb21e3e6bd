"""Host side of the two native kernels of the plane-mask head: the call signatures of the reference's
`maskrcnn_benchmark.layers.nms` (layers/nms.py:3-7 -> `_C.nms(boxes, scores, threshold)`) and `ROIAlign` (layers/roi_align.py:8-60),
on device tensors, executed by csrc/detector.hip.  No CPU fallback."""
import torch

from . import _lib as L


def nms(boxes, scores, nms_thresh, inclusive=False):
    """boxes (N,4) xyxy fp32, scores (N,) on the GPU -> kept indices (int64, ascending), like `_C.nms`.
    inclusive=False is the reference's CUDA rule (IoU > thresh, csrc/cuda/nms.cu:57), True its CPU rule (>=, csrc/cpu/nms_cpu.cpp:60)."""
    if not (boxes.is_cuda and scores.is_cuda):
        raise RuntimeError("nms runs on the GPU only (no CPU fallback)")
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    b = boxes.contiguous().float()
    order = torch.sort(scores.float(), descending=True, stable=True)[1].to(torch.int32)      # plumbing: the reference sorts with torch too
    keep = torch.empty(n, dtype=torch.int32, device=b.device)
    n_keep = torch.zeros(1, dtype=torch.int32, device=b.device)
    scratch = torch.empty(L.lib().vidc_nms_scratch_bytes(n), dtype=torch.uint8, device=b.device)
    L.check(L.lib().vidc_nms(L.ptr(b), L.ptr(order), n, float(nms_thresh), int(inclusive), L.ptr(keep), L.ptr(n_keep), L.ptr(scratch),
                             L.current_stream()), "nms")
    return keep[: int(n_keep.item())].long()


def roi_align_nhwc(x_nhwc, rois, output_size, spatial_scale, sampling_ratio):
    """x (N,H,W,C) fp32 NHWC, rois (K,5) = (batch index, x1, y1, x2, y2) -> (K, PH, PW, C)."""
    if not (x_nhwc.is_cuda and rois.is_cuda):
        raise RuntimeError("roi_align runs on the GPU only (no CPU fallback)")
    x, r = x_nhwc.contiguous().float(), rois.contiguous().float()
    N, H, W, Cc = x.shape
    K = r.shape[0]
    PH, PW = output_size
    y = torch.empty((K, PH, PW, Cc), dtype=torch.float32, device=x.device)
    L.check(L.lib().vidc_roi_align_forward(L.ptr(x), L.ptr(r), L.ptr(y), K, Cc, H, W, Cc, PH, PW, float(spatial_scale), int(sampling_ratio),
                                           L.current_stream()), "roi_align")
    return y


def roi_align(x_nchw, rois, output_size, spatial_scale, sampling_ratio):
    """The reference's layout: (N,C,H,W) in, (K,C,PH,PW) out (layers/roi_align.py:8-60); permutes around the NHWC kernel."""
    return roi_align_nhwc(x_nchw.permute(0, 2, 3, 1), rois, output_size, spatial_scale, sampling_ratio).permute(0, 3, 1, 2).contiguous()
