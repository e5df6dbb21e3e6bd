"""Times the few-row grouped GEMMs of the Winograd layers (M = 80 at program batch 4) on the general tiles and on the streamed tile
(csrc/wgemm.hip), weights HBM-cold (a rotation of weight copies larger than the caches), back-to-back launches between two HIP events.

    python tools/wgemm_bench.py [--iters 40]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from vi_depth_completion_amd import _lib as L

torch.set_grad_enabled(False)
DEV = "cuda"
SHAPES = [(80, 144, 256, 256, "layer3 x22"), (80, 64, 512, 512, "layer4 F2 x2"), (80, 72, 512, 512, "sn/feature3_upsamping.3"), (80, 72, 1536, 1536, "dc/feature3_upsamping.3"),
          (80, 16, 3072, 3072, "dc/feature4_upsamping.3 F2"), (80, 16, 1024, 1024, "sn/feature4_upsamping.3 F2"), (40, 144, 256, 256, "layer3 at F = 2"), (20, 144, 256, 256, "layer3 at F = 1")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--tiles", default="28,6,10,21,32,4,40,41")
    ap.add_argument("--chunks", default="0,16,32,48,64,96,128")
    a = ap.parse_args()
    lib = L.lib()
    tiles = [int(t) for t in a.tiles.split(",")]
    for (M, G, K, N, label) in SHAPES:
        wbytes = G * N * K * 4
        copies = max(2, min(24, (1 << 30) // wbytes))
        x = torch.randn(1, 1, M, G * K, device=DEV)
        ws = [torch.randn(G, N, K, device=DEV) * 0.05 for _ in range(copies)]
        y = torch.empty(1, 1, M, G * N, device=DEV)
        one, zero = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
        flop = 2.0 * M * N * K * G
        res = []
        for t in tiles:
            for ch in ([int(c) for c in a.chunks.split(",")] if t >= 40 else [1, 2]):
                d = L.ConvDesc()
                d.x, d.y, d.scale1, d.shift1 = L.ptr(x), L.ptr(y), L.ptr(one), L.ptr(zero)
                d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 1, 1, M, K, G * K, 1, M, N, G * N
                d.KH, d.KW, d.stride, d.pad, d.groups, d.flags = 1, 1, 1, 0, G, 0
                d.x_gs, d.w_gs, d.y_gs, d.p_gs = K, N * K, N, 0
                d.tile, d.splitk, d.precision = t, max(ch, 1) if t >= 40 else ch, 0
                wsb = None
                need = lib.vidc_conv2d_workspace_bytes(C.byref(d))
                if need:
                    wsb = torch.zeros(need // 4, device=DEV)
                    d.workspace = L.ptr(wsb)
                if t >= 40:
                    d.splitk = ch

                def go(i):
                    d.w = L.ptr(ws[i % copies])
                    return lib.vidc_conv2d_bn_act(C.byref(d), L.current_stream())
                if go(0) != 0:
                    continue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.iters):
                    go(i + 1)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                res.append((us, "%s%s" % (L.TILE_NAMES[t], (":c%d" % ch) if t >= 40 else (":sk%d" % ch))))
        res.sort()
        print("M%d_N%d_K%d_G%d  %-28s %6.2f GFLOP  weights %5.1f MB | " % (M, N, K, G, label, flop / 1e9, wbytes / 1e6) +
              "  ".join("%s %.1f us (%.0f TF)" % (n, us, flop / us / 1e6) for us, n in res[:7]), flush=True)


if __name__ == "__main__":
    main()
