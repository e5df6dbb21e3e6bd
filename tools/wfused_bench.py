"""Times the small-map 3x3 layers as Winograd F(4x4) in ONE launch (csrc/wfused.hip) against the three-launch path (input transform + grouped GEMM on the
table's tile + output transform), weights HBM-cold (a rotation of weight copies), back-to-back launches between two HIP events.

    python tools/wfused_bench.py [--iters 40]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from vi_depth_completion_amd import _lib as L
from vi_depth_completion_amd import ops

torch.set_grad_enabled(False)
DEV = "cuda"
# (B, H, W, cin, cout, G, label, tile of the three-launch GEMM in the measured table)
SHAPES = [(4, 16, 20, 256, 256, 4, "layer3 conv2 x22", 28), (4, 32, 40, 128, 128, 4, "layer2 conv2 x3", 29), (4, 16, 20, 512, 512, 2, "sn/feature3_upsamping.3", 28),
          (4, 32, 40, 256, 256, 3, "sn/feature2_upsamping.3", 29), (4, 16, 20, 256, 256, 1, "layer3, one group (head tick)", 28), (4, 16, 20, 256, 256, 3, "layer3, three groups (tail tick)", 28)]


def timed(fn, iters):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i + 1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    for (B, H, W, cin, cout, G, label, tile) in SHAPES:
        x = torch.randn(B, H, W, G * cin, device=DEV)
        copies = max(2, min(16, (1 << 29) // (G * 36 * cout * cin * 4)))
        us = [torch.randn(G * 36, cout, cin, device=DEV) * 0.05 for _ in range(copies)]      # (random values: the fused launch reads them in its packed order, the GEMM as [pos][Cout][Cin])
        s1, b1 = torch.rand(G, cout, device=DEV) + 0.5, torch.randn(G, cout, device=DEV) * 0.1
        t_f = timed(lambda i: ops.conv3x3_winograd_fused(x, None, s1, b1, relu1=True, u=us[i % copies]), a.iters)
        # the three launches separately (the python wrappers allocate; time each kernel family on its own)
        v = ops.winograd_input_transform(x, cin, 4)
        t_in = timed(lambda i: ops.winograd_input_transform(x, cin, 4), a.iters)
        tiles = v.shape[0]
        one, zero = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
        t_g = timed(lambda i: ops.conv2d_bn_act(v.view(1, 1, tiles, -1), us[i % copies].view(G * 36, cout, cin), one, zero, 1, 1, tile=tile, groups=G * 36), a.iters)
        mm = torch.randn(tiles, 36 * G * cout, device=DEV)
        t_out = timed(lambda i: ops.winograd_output_transform(mm, B, H, W, cout, 4, s1, b1, relu1=True), a.iters)
        flop = 2.0 * tiles * 36 * cin * cout * G
        print("%-34s B%d %dx%d %d->%d G%d  %5.2f GFLOP | fused %6.1f us (%5.1f TF) | three launches %5.1f + %5.1f + %5.1f = %6.1f us" %
              (label, B, H, W, cin, cout, G, flop / 1e9, t_f, flop / t_f / 1e6, t_in, t_g, t_out, t_in + t_g + t_out), flush=True)


if __name__ == "__main__":
    main()
