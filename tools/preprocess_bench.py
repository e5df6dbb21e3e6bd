#!/usr/bin/env python3
"""HBM roofline of the device-side frame pre-processing (csrc/preprocess.hip): PIL-exact bilinear resize + ToTensor.
Algorithmic bytes per frame: source H*W*3 (uint8) in + 3*Ho*Wo*4 (fp32 CHW) out.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vi_depth_completion_amd.preprocess import FramePreprocessor      # noqa: E402


def main():
    for (h, w) in ((480, 640), (720, 1280)):
        pre = FramePreprocessor("cuda", in_hw=(h, w), out_hw=(240, 320))
        for B in (1, 8, 32):
            x = torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device="cuda")
            for _ in range(3):
                pre.resize(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                pre.resize(x)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            nbytes = B * (h * w * 3 + 3 * 240 * 320 * 4)
            print("resize %4dx%-4d -> 320x240  batch %2d: %7.1f us  %6.1f GB/s algorithmic (%.2f MB)  %.0f frames/s" % (
                w, h, B, us, nbytes / us / 1e3, nbytes / 1e6, B / us * 1e6))


if __name__ == "__main__":
    main()
