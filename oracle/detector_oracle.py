"""CPU restatement of the two native kernels the plane-mask head reaches: TEST INFRASTRUCTURE (imported only
by tests/).

* `nms`: follows plane_mask_detection/maskrcnn_benchmark/csrc/cpu/nms_cpu.cpp:5-67 (greedy NMS in descending score order, boxes
  (x1, y1, x2, y2) with the "+1" area convention, suppress when IoU >= threshold) and, with `strict=True`, the CUDA variant
  csrc/cuda/nms.cu:13-21,23-63 (IoU > threshold).  Pinned against the reference's own known-answer vectors
  (plane_mask_detection/tests/test_nms.py:11-54 and :60-255 -> tests/golden/nms_reference_vectors.npz, extracted as DATA by
  oracle/tools/make_golden_detector.py).
* `roi_align_forward`: follows csrc/cpu/ROIAlign_cpu.cpp:17-218 (Caffe2 ROIAlign, no half-pixel shift, malformed ROIs forced to 1x1,
  sampling_ratio <= 0 -> ceil(roi / pooled) samples per bin).  PARITY UNPINNED: the reference has no test or fixture for ROIAlign, and
  its C++ sources do not build against this container's PyTorch 2.10 headers (`AT_DISPATCH_FLOATING_TYPES(dets.type(), ...)` no longer
  converts; tried with g++ on csrc/cpu/*.cpp directly), so there is no oracle/_ref build either.
"""
import math

import numpy as np


def nms(boxes, scores, threshold, strict=False):
    """boxes (N,4) float32, scores (N,) -> kept indices, ascending (like `at::nonzero(suppressed == 0)` / the sorted CUDA result)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    suppressed = np.zeros(n, bool)
    thr = np.float32(threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            w = max(np.float32(0), min(x2[i], x2[j]) - max(x1[i], x1[j]) + np.float32(1))
            h = max(np.float32(0), min(y2[i], y2[j]) - max(y1[i], y1[j]) + np.float32(1))
            inter = np.float32(w * h)
            ovr = inter / np.float32(areas[i] + areas[j] - inter)
            if (ovr > thr) if strict else (ovr >= thr):
                suppressed[j] = True
    return np.flatnonzero(~suppressed).astype(np.int64)


def roi_align_forward(inp_nchw, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio):
    """inp (N,C,H,W) float32, rois (K,5) = (batch index, x1, y1, x2, y2) -> (K,C,pooled_h,pooled_w) float32."""
    inp = np.asarray(inp_nchw, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    _, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.zeros((K, C, pooled_h, pooled_w), np.float32)
    f = np.float32
    for n in range(K):
        b = int(rois[n, 0])
        sw, sh, ew, eh = (f(rois[n, 1] * f(spatial_scale)), f(rois[n, 2] * f(spatial_scale)), f(rois[n, 3] * f(spatial_scale)),
                          f(rois[n, 4] * f(spatial_scale)))
        rw, rh = max(f(ew - sw), f(1)), max(f(eh - sh), f(1))
        bh, bw = f(rh / f(pooled_h)), f(rw / f(pooled_w))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / pooled_h))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pooled_w))
        count = f(gh * gw)
        for ph in range(pooled_h):
            for pw in range(pooled_w):
                acc = np.zeros(C, np.float32)
                for iy in range(gh):
                    y = f(sh + f(ph) * bh + f(f(iy + 0.5) * bh / f(gh)))
                    for ix in range(gw):
                        x = f(sw + f(pw) * bw + f(f(ix + 0.5) * bw / f(gw)))
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yy, xx = max(y, f(0)), max(x, f(0))
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = f(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xx = f(xl)
                        else:
                            xh = xl + 1
                        ly, lx = f(yy - f(yl)), f(xx - f(xl))
                        hy, hx = f(f(1) - ly), f(f(1) - lx)
                        acc = acc + (f(hy * hx) * inp[b, :, yl, xl] + f(hy * lx) * inp[b, :, yl, xh] + f(ly * hx) * inp[b, :, yh, xl] +
                                     f(ly * lx) * inp[b, :, yh, xh]).astype(np.float32)
                out[n, :, ph, pw] = acc / count
    return out
