#!/usr/bin/env python3
"""HBM roofline of the bandwidth-bound kernels of the hot path (north_star: "rocprof-reported HBM GB/s for the warp/upsample
kernels"): GPU-bound timing (captured hipGraph of 20 back-to-back launches through libvidc's program API would need op descriptors;
here: 50 launches between two HIP events after a warm-up, direct C-ABI calls) at batch 1 (the bench configuration: launch-bound) and
at batch 8 / 32 (where the kernels are bandwidth-bound).  Algorithmic bytes per SURVEY §8d.  GPU only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vi_depth_completion_amd import _lib as L, synthetic as S      # noqa: E402
from vi_depth_completion_amd.networks.warping_2dof_alignment import Warping2DOFAlignment      # noqa: E402

PEAK = 8000.0       # GB/s, MI355X_MICROARCH.md


def timed(fn, n=50):
    for _ in range(5):
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
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    H, W = 256, 320
    rows = []
    for B in (1, 4, 8, 32):
        wp = Warping2DOFAlignment(202.0, 202.0, 0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
        b = S.synthetic_batch(B, H, W, 1234)
        x = b["image"].cuda()
        g, a = b["gravity"].cuda(), b["aligned_direction"].cuda()
        p = wp._params(g, a)
        y = torch.empty_like(x)
        us = timed(lambda: lib.vidc_warp2dof_fwd(x.data_ptr(), p.data_ptr(), y.data_ptr(), B, 3, H, W, wp.cx, wp.cy, 0, st))
        rows.append(("warp_fwd 3ch %dx%d" % (W, H), B, us, 2 * x.numel() * 4))
        us = timed(lambda: lib.vidc_warp2dof_inv_rot_norm(x.data_ptr(), p.data_ptr(), y.data_ptr(), B, H, W, wp.cx, wp.cy, 0, 1, st))
        rows.append(("warp_inv_rot_norm", B, us, 2 * x.numel() * 4))
        # the surface-normal stem gathering its input through the forward warp (round 5; replaces warp_fwd + stem in the programs):
        # algorithmic bytes = the image read once + the 64-channel half-resolution output
        wt = torch.randn(64, 3, 3, 3, device="cuda") * 0.2
        ys = torch.empty(B, H // 2, W // 2, 64, device="cuda")
        us = timed(lambda: lib.vidc_stem_conv3x3s2_warped(x.data_ptr(), p.data_ptr(), wt.data_ptr(), ys.data_ptr(), B, H, W, 64, 64, 1, None, 0, wp.cx, wp.cy, 0, st))
        rows.append(("stem conv through the warp", B, us, (x.numel() + ys.numel()) * 4))
        us = timed(lambda: lib.vidc_stem_conv3x3s2(x.data_ptr(), wt.data_ptr(), ys.data_ptr(), B, 3, H, W, 64, 64, 1, None, 0, st))
        rows.append(("stem conv (plain input)", B, us, (x.numel() + ys.numel()) * 4))
        for (h, w, Cc, Ho, Wo) in ((32, 40, 768, 64, 80), (16, 20, 1536, 32, 40), (32, 40, 256, 64, 80)):
            xi = torch.randn(B, h, w, Cc, device="cuda")
            yo = torch.empty(B, Ho, Wo, Cc, device="cuda")
            us = timed(lambda: lib.vidc_upsample_bilinear_ac(xi.data_ptr(), yo.data_ptr(), B, h, w, Cc, Cc, Ho, Wo, Cc, 0, None, st))
            rows.append(("upsample %dx%dx%d -> %dx%d" % (h, w, Cc, Ho, Wo), B, us, (xi.numel() + yo.numel()) * 4))
        # Winograd transforms (csrc/winograd.hip) of the layers that dominate them: dc/feature1_upsamping.0 (768 ch at 64x80, F(4x4)) and the
        # level-2 launch of the depth decoder (3 x 768 ch at 32x40, F(4x4)); algorithmic bytes = the NHWC tensor + the transform-domain tensor
        for (h, w, cin, G, m) in ((64, 80, 768, 1, 4), (32, 40, 768, 3, 4), (16, 20, 256, 4, 4), (16, 20, 1536, 2, 2)):
            a2 = (m + 2) ** 2
            tiles = B * (-(-h // m)) * (-(-w // m))
            xi = torch.randn(B, h, w, G * cin, device="cuda")
            v = torch.empty(tiles, a2 * G * cin, device="cuda")
            one = torch.ones(G * cin, device="cuda")
            us = timed(lambda: lib.vidc_winograd_input_transform(xi.data_ptr(), v.data_ptr(), B, h, w, G * cin, G * cin, cin, m, 0, 0, st))
            rows.append(("wino_in F%d %dx%dx%d" % (m, h, w, G * cin), B, us, (xi.numel() + v.numel()) * 4))
            us = timed(lambda: lib.vidc_winograd_output_transform(v.data_ptr(), xi.data_ptr(), None, one.data_ptr(), one.data_ptr(), None, None, B, h, w, G * cin, cin, G * cin, m,
                                                                  L.RELU1, 0, st))
            rows.append(("wino_out F%d %dx%dx%d" % (m, h, w, G * cin), B, us, (xi.numel() + v.numel()) * 4))
        xi = torch.randn(B, 128, 160, 512, device="cuda")
        yo = torch.empty(B, 64, 80, 512, device="cuda")
        us = timed(lambda: lib.vidc_maxpool3x3s2(xi.data_ptr(), yo.data_ptr(), B, 128, 160, 512, 512, 512, None, st))
        rows.append(("maxpool 128x160x512 (4 pyramids)", B, us, (xi.numel() + yo.numel()) * 4))
    print("%-36s %5s %10s %10s %10s %8s" % ("kernel", "batch", "us", "MB", "GB/s", "of peak"))
    for name, B, us, nbytes in rows:
        print("%-36s %5d %10.1f %10.2f %10.1f %7.1f%%" % (name, B, us, nbytes / 1e6, nbytes / us / 1e3, 100 * nbytes / us / 1e3 / PEAK))


if __name__ == "__main__":
    main()
