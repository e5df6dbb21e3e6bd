#!/usr/bin/env python3
"""Times the plane-mask head kernels (csrc/detector.hip) at Mask R-CNN-like sizes.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vi_depth_completion_amd import _lib as L, detector      # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    rng = np.random.RandomState(0)
    lib = L.lib()
    for n in (1000, 2000, 4000):
        c = rng.uniform(0, 320, (n, 2))
        wh = rng.uniform(8, 120, (n, 2))
        boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)).cuda()
        order = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
        keep = torch.empty(n, dtype=torch.int32, device="cuda")
        nk = torch.zeros(1, dtype=torch.int32, device="cuda")
        scratch = torch.empty(lib.vidc_nms_scratch_bytes(n), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        us = timed(lambda: lib.vidc_nms(boxes.data_ptr(), order.data_ptr(), n, 0.7, 0, keep.data_ptr(), nk.data_ptr(), scratch.data_ptr(), st))
        print("nms n=%5d thresh 0.7: %7.1f us (mask + on-device greedy reduction + compaction), kept %d; the reference copies a %d KiB mask to the host instead"
              % (n, us, int(nk.item()), n * ((n + 63) // 64) * 8 // 1024))
    for (K, C, P, H, W) in ((1000, 256, 7, 60, 80), (100, 256, 14, 60, 80)):
        x = torch.randn(1, H, W, C, device="cuda")
        x1, y1 = rng.uniform(0, 250, K), rng.uniform(0, 180, K)
        rois = torch.from_numpy(np.stack([np.zeros(K), x1, y1, x1 + rng.uniform(10, 120, K), y1 + rng.uniform(10, 100, K)], 1).astype(np.float32)).cuda()
        us = timed(lambda: detector.roi_align_nhwc(x, rois, (P, P), 0.25, 2))
        out_bytes = K * P * P * C * 4
        print("roi_align K=%4d C=%d %dx%d (sampling 2): %7.1f us, %.1f MB written -> %.0f GB/s of output" % (K, C, P, P, us, out_bytes / 1e6, out_bytes / us / 1e3))


if __name__ == "__main__":
    main()
