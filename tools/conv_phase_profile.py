#!/usr/bin/env python3
"""Debug tool: in-kernel phase profile of the fused conv (needs libvidc_timing.so = conv_mfma.hip built with
-DVIDC_CONV_TIMING).  Prints, per workgroup-averaged, the shader-clock cycles from kernel entry to: setup done,
prologue DMAs issued, first stage landed, main loop done, epilogue issued; plus start skew / end from the 100 MHz clock."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vi_depth_completion_amd import _lib as L          # noqa: E402
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ.get("VIDC_TIMING_LIB", "libvidc_timing.so"))
from vi_depth_completion_amd import synthetic as S     # noqa: E402

SHAPES = {  # name: (H, W, cin, cout, k, G, tile, precision)
    "l3_1x1_1024to256_G4": (16, 20, 1024, 256, 1, 4, 10, 1),
    "l3_3x3_G4": (16, 20, 256, 256, 3, 4, 10, 1),
    "l3_3x3_G4_L": (16, 20, 256, 256, 3, 4, 15, 1),
    "l3_1x1_1024to256_G4_L": (16, 20, 1024, 256, 1, 4, 15, 1),
    "l3_1x1_256to1024_G4_L": (16, 20, 256, 1024, 1, 4, 17, 1),
    "f1_3x3_768_L": (64, 80, 768, 768, 3, 1, 19, 1),
    "l3_1x1_256to1024_G4": (16, 20, 256, 1024, 1, 4, 4, 1),
    "l3_1x1_256to1024_G1": (16, 20, 256, 1024, 1, 1, 6, 1),
    "f1_3x3_768": (64, 80, 768, 768, 3, 1, 3, 1),
    "now_l3_1x1_256to1024_G4": (16, 20, 256, 1024, 1, 4, 20, 1),
    "now_l3_1x1_1024to256_G4": (16, 20, 1024, 256, 1, 4, 23, 1),
    "now_l3_3x3_G4_64x64k2d4": (16, 20, 256, 256, 3, 4, 13, 1),
    "f1_3x3_768_128x128d3L": (64, 80, 768, 768, 3, 1, 25, 1),
    "f1_3x3_768_128x128d3": (64, 80, 768, 768, 3, 1, 24, 1),
    # round 4: two stream items per launch (M = 640), the fp32 leg's tilings (split-K 1 here: the stamps live in the split-K workspace)
    "b2_l3_1x1_256to1024_G4_fp32": (32, 20, 256, 1024, 1, 4, 29, 0),
    "b2_l3_1x1_1024to256_G4_fp32": (32, 20, 1024, 256, 1, 4, 32, 0),
    "b2_l3_3x3_G4_fp32": (32, 20, 256, 256, 3, 4, 28, 0),
    # round 5: four stream items per launch (M = 1280), fp32 leg
    "b4_l3_1x1_1024to256_G4_fp32": (64, 20, 1024, 256, 1, 4, 32, 0),
    "b4_l3_1x1_1024to256_G4_fp32_32x32k4d2": (64, 20, 1024, 256, 1, 4, 30, 0),
    "b4_l3_1x1_1024to256_G4_fp32_64x64k2": (64, 20, 1024, 256, 1, 4, 5, 0),
    "b4_l3_1x1_256to1024_G4_fp32": (64, 20, 256, 1024, 1, 4, 35, 0),
    "b4_l3_1x1_256to1024_G4_fp32_64x64d2": (64, 20, 256, 1024, 1, 4, 29, 0),
    "b4_l3_wino4_gemm_fp32": (4, 20, 256, 256, 1, 144, 8, 0),
}


def main():
    dev = "cuda"
    lib = L.lib()
    only = os.environ.get("VIDC_PHASE_ONLY", "")
    for name, (H, W, cin, cout, k, G, tile, prec) in SHAPES.items():
        if only and only not in name:
            continue
        x = S.normal01(1, "x", (1, H, W, G * cin)).float().to(dev)
        w = S.normal01(1, "w", (G, cout, k * k * cin), scale=0.05).float().to(dev)
        s1, b1 = torch.ones(G, cout, device=dev), torch.zeros(G, cout, device=dev)
        y = torch.empty(1, H, W, G * cout, device=dev)
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale1, d.shift1 = x.data_ptr(), w.data_ptr(), y.data_ptr(), s1.data_ptr(), b1.data_ptr()
        d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 1, H, W, cin, G * cin, H, W, cout, G * cout
        d.KH, d.KW, d.stride, d.pad, d.flags, d.groups = k, k, 1, k // 2, L.RELU1, G
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, cout * k * k * cin, cout, cout
        d.tile, d.splitk, d.precision = tile, 1, prec
        dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
        d.workspace = dbg.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        for rep in range(3):
            junk.fill_(rep)                    # evict L2 / Infinity Cache: weights must come from HBM like in the real frame
            torch.cuda.synchronize()
            dbg.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "conv")
            e1.record()
            torch.cuda.synchronize()
            t = dbg.cpu().numpy().reshape(-1, 16)
            nb = int((t[:, 7] != 0).sum())
            t = t[:nb]
            start = (t[:, 6] - t[:, 6].min()) / 100.0          # us, 100 MHz clock
            end = (t[:, 7] - t[:, 6].min()) / 100.0
            dur = end - start
            order = np.argsort(start)
            late = int((start > 0.5 * np.median(dur)).sum())       # workgroups that started after others had run for a while (a later round)
            print("    per-workgroup: duration median %.2f us (min %.2f, max %.2f); %d of %d started late (> half a median duration after the first); "
                  "sum of durations / (256 CUs x last end) = %.2f workgroups resident per CU on average" % (
                      np.median(dur), dur.min(), dur.max(), late, nb, dur.sum() / (256.0 * end.max())))
            print("%-22s tile %-8s wgs %4d  event %.1f us | start skew max %.2f us, last end %.2f us | cycles: setup %.0f  issued %.0f  "
                  "first-data %.0f  loop-done %.0f  end %.0f | sub: args %.0f  B-issued %.0f  A-decoded %.0f  affine-loads %.0f" % (name, L.TILE_NAMES[tile], nb, e0.elapsed_time(e1) * 1e3, start.max(), end.max(),
                                                                  t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(), t[:, 4].mean(),
                                                                  t[:, 8].mean(), t[:, 9].mean(), t[:, 10].mean(), t[:, 11].mean()))


if __name__ == "__main__":
    main()
