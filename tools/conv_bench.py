#!/usr/bin/env python3
"""Times the fused conv kernel on the layer shapes of the two networks (GPU only).
    python tools/conv_bench.py [--tiles 1,4,7] [--shapes all|key] [--check]
Prints per (shape, tile, splitk): microseconds per launch (hipEvents, 20 back-to-back launches) and TFLOP/s."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vi_depth_completion_amd import _lib as L          # noqa: E402
from vi_depth_completion_amd import synthetic as S     # noqa: E402

# (B*Ho*Wo as H x W), Cin, Cout, k, stride, groups, count per frame
KEY_SHAPES = [
    ((64, 80), 768, 768, 3, 1, 1, 1),      # dc f1 3x3: 54 GF
    ((32, 40), 768, 768, 3, 1, 1, 3),
    ((16, 20), 1536, 1536, 3, 1, 1, 2),
    ((8, 10), 3072, 3072, 3, 1, 1, 1),
    ((16, 20), 256, 256, 3, 1, 3, 22),     # layer3 3x3 grouped
    ((16, 20), 256, 1024, 1, 1, 3, 23),
    ((16, 20), 1024, 256, 1, 1, 3, 22),
    ((16, 20), 256, 256, 3, 1, 1, 22),     # SN layer3
    ((16, 20), 256, 1024, 1, 1, 1, 23),
    ((16, 20), 1024, 256, 1, 1, 1, 22),
    ((64, 80), 768, 384, 1, 1, 1, 4),
    ((64, 80), 64, 256, 1, 1, 3, 3),
    ((128, 160), 64, 128, 3, 1, 3, 1),
    ((64, 80), 64, 64, 3, 1, 3, 3),
    ((32, 40), 128, 128, 3, 1, 3, 3),
    ((8, 10), 512, 512, 3, 1, 3, 2),
    ((8, 10), 512, 2048, 1, 1, 3, 3),
    ((16, 20), 256, 256, 3, 1, 4, 22),     # joint 4-pyramid launch (pipeline.run_stream software pipelining)
    ((16, 20), 256, 1024, 1, 1, 4, 23),
    ((16, 20), 1024, 256, 1, 1, 4, 22),
    ((512, 80), 768, 768, 3, 1, 1, 1),     # [20] the 54 GF decoder layer at batch 8 (M = 40 960)
    ((128, 80), 768, 768, 3, 1, 1, 6),     # [21] the batch-8 M = 9600-class decoder layers (M = 10 240)
]


def run(shape, tile, splitk, iters=20, check=False, precision=0):
    (H, W), cin, cout, k, stride, G, _ = shape
    dev = "cuda"
    x = S.normal01(1, "cb.x", (1, H, W, G * cin)).float().to(dev)
    w = S.normal01(1, "cb.w", (G, cout, k * k * cin), scale=0.05).float().to(dev)
    s1 = torch.ones(G, cout, device=dev)
    b1 = torch.zeros(G, cout, device=dev)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(1, Ho, Wo, G * cout, device=dev)
    d = L.ConvDesc()
    d.x, d.w, d.y, d.scale1, d.shift1 = x.data_ptr(), w.data_ptr(), y.data_ptr(), s1.data_ptr(), b1.data_ptr()
    d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 1, H, W, cin, G * cin, Ho, Wo, cout, G * cout
    d.KH, d.KW, d.stride, d.pad, d.flags, d.groups = k, k, stride, pad, L.RELU1, G
    d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, cout * k * k * cin, cout, cout
    d.tile, d.splitk, d.precision = tile, splitk, precision
    lib = L.lib()
    if tile == 0:
        L.check(lib.vidc_conv2d_plan(C.byref(d)), "plan")
    ws = torch.zeros(max(4, lib.vidc_conv2d_workspace_bytes(C.byref(d)) // 4), device=dev)
    d.workspace = ws.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.vidc_conv2d_bn_act(C.byref(d), st)
    if rc != 0:
        return None, d.tile, d.splitk, lib.vidc_last_error().decode()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.vidc_conv2d_bn_act(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    err = None
    if check:
        import torch.nn.functional as F
        xc = x.cpu().permute(0, 3, 1, 2)
        outs = []
        for g in range(G):
            wg = w[g].cpu().view(cout, cin // 32, k, k, 32).permute(0, 1, 4, 2, 3).reshape(cout, cin, k, k)
            outs.append(F.relu(F.conv2d(xc[:, g * cin:(g + 1) * cin], wg, None, stride, pad)))
        ref = torch.cat(outs, 1).permute(0, 2, 3, 1)
        err = (y.cpu() - ref).abs().max().item()
    return us, d.tile, d.splitk, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--splitk", default="1")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--precision", type=int, default=0, help="0 fp32, 1 bf16x3 (timing only: operands are not re-split)")
    ap.add_argument("--shapes", default="key")
    ap.add_argument("--only", type=int, default=-1, help="index into KEY_SHAPES")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    sks = [int(t) for t in a.splitk.split(",")]
    total = {}
    for si, sh in enumerate(KEY_SHAPES):
        if a.only >= 0 and si != a.only:
            continue
        (H, W), cin, cout, k, stride, G, cnt = sh
        fl = 2.0 * H * W * cout * cin * k * k * G / (stride * stride)
        best = None
        for t in tiles:
            for sk in sks:
                us, tt, ss, err = run(sh, t, sk, iters=a.iters, check=a.check, precision=a.precision)
                if us is None:
                    print("  M%-6d N%-5d K%-6d G%d  tile %-8s sk%-2d  FAILED %s" % (H * W, cout, cin * k * k, G, L.TILE_NAMES.get(tt, tt), ss, err))
                    continue
                print("  M%-6d N%-5d K%-6d G%d  tile %-8s sk%-2d  %8.1f us  %6.1f TF/s%s" % (
                    H * W, cout, cin * k * k, G, L.TILE_NAMES.get(tt, tt), ss, us, fl / us / 1e6, "" if err is None else "  err %.1e" % err))
                if best is None or us < best[0]:
                    best = (us, tt, ss)
        if best:
            total[sh] = best[0] * cnt
            print("  -> best %.1f us (tile %s sk%d) x%d = %.0f us/frame" % (best[0], L.TILE_NAMES.get(best[1], best[1]), best[2], cnt, best[0] * cnt))
    print("sum over key shapes: %.0f us/frame" % sum(total.values()))


if __name__ == "__main__":
    main()
