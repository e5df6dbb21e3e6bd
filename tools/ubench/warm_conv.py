#!/usr/bin/env python3
"""Experiment: how much of a small conv's time is cold-weight latency?  Times the layer-3 shapes (4 groups) with weights
(a) HBM-cold (512 MB of junk written in between), (b) warmed by another kernel reading them once (w.sum(): L2 of whatever XCDs
ran that kernel + Infinity Cache), (c) hot (same launch repeated)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vi_depth_completion_amd import _lib as L, synthetic as S, engine      # noqa: E402

SHAPES = {"l3_1x1_1024to256_G4": (16, 20, 1024, 256, 1, 4), "l3_3x3_G4": (16, 20, 256, 256, 3, 4), "l3_1x1_256to1024_G4": (16, 20, 256, 1024, 1, 4)}


def main():
    dev = "cuda"
    lib = L.lib()
    tab = engine.tuning_table()
    for name, (H, W, cin, cout, k, G) in SHAPES.items():
        sig = "M%d_N%d_K%d_k%ds1_G%d" % (H * W, cout, k * k * cin, k, G)
        ent = tab.get(sig, [6, 1, 1])
        x = S.normal01(1, "x", (1, H, W, G * cin)).float().to(dev)
        w = S.normal01(1, "w", (G, cout, k * k * cin), scale=0.05).float().to(dev)
        s1, b1 = torch.ones(G, cout, device=dev), torch.zeros(G, cout, device=dev)
        y = torch.empty(1, H, W, G * cout, device=dev)
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale1, d.shift1 = x.data_ptr(), w.data_ptr(), y.data_ptr(), s1.data_ptr(), b1.data_ptr()
        d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 1, H, W, cin, G * cin, H, W, cout, G * cout
        d.KH, d.KW, d.stride, d.pad, d.flags, d.groups = k, k, 1, k // 2, L.RELU1, G
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, cout * k * k * cin, cout, cout
        d.tile, d.splitk, d.precision = ent[0], ent[1], ent[2]
        st = torch.cuda.current_stream().cuda_stream
        junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        res = {}
        for mode in ("cold", "warm", "hot"):
            ts = []
            for rep in range(5):
                junk.fill_(rep)
                if mode == "warm":
                    _ = w.sum(); _ = x.sum()
                if mode == "hot":
                    L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "conv")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "conv")
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res[mode] = sorted(ts)[len(ts) // 2]
        print("%-22s tile %-8s  cold %.1f us   warm(MALL/L2 by another kernel) %.1f us   hot %.1f us" % (name, L.TILE_NAMES[ent[0]], res["cold"], res["warm"], res["hot"]))


if __name__ == "__main__":
    main()
