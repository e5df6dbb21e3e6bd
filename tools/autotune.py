#!/usr/bin/env python3
"""Measures every unique fused-conv signature of the two networks on the GPU and writes the per-shape
(tile, splitk) table `vi_depth_completion_amd/conv_tuning.json` that engine.Program consults.

    python tools/autotune.py --heights 240,256 --batches 1

Each candidate is timed as 10 back-to-back launches (HIP events, min of 2 repeats) on the program's real buffers.
All tilings compute the same fp32 sums in a different association order, so results differ by rounding only;
the parity tests run with whatever table is committed.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import _lib as L, engine                       # noqa: E402
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN   # noqa: E402
from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction   # noqa: E402

OUT = os.path.join(ROOT, "vi_depth_completion_amd", "conv_tuning.json")
TILE_DIMS = {1: (128, 128), 2: (128, 64), 3: (64, 128), 4: (64, 64), 5: (64, 64), 6: (32, 64), 7: (32, 32), 8: (32, 128), 9: (32, 32)}


def time_desc(lib, d, st, iters=10):
    best = 1e30
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            if lib.vidc_conv2d_bn_act(C.byref(d), st) != 0:
                return None
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heights", default="240,256")
    ap.add_argument("--batches", default="1")
    ap.add_argument("--splitk", default="1,2,4,8,16")
    a = ap.parse_args()
    dev = torch.device("cuda")
    lib = L.lib()
    engine._TUNING = {}                      # measure against the cost-model plan, not an older table
    table, report = {}, []
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)     # 256 MB split-K scratch
    for H in [int(v) for v in a.heights.split(",")]:
        for B in [int(v) for v in a.batches.split(",")]:
            cc = np.array([0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0])
            sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0]), cc_img=cc).to(dev).eval()
            dc = ModifiedFPN().to(dev).eval()
            for prog in (sn.program(B, dev), dc.program(B, H, 320, dev)):
                for op, name in zip(prog.c_ops, prog.op_names):
                    if op.kind != L.OP_CONV:
                        continue
                    sig = name.split(" ")[1]
                    if sig in table:
                        continue
                    d = L.ConvDesc.from_buffer_copy(op.u.conv)
                    d.workspace = ws.data_ptr()
                    d.flags &= ~L.ACCUM                     # timing launches must not accumulate into live data forever
                    base = (d.tile, d.splitk)
                    M = d.B * d.Ho * d.Wo
                    cands = []
                    for t, (bm, bn) in TILE_DIMS.items():
                        if bn > max(64, d.Cout) or bm >= 4 * max(32, M):
                            continue
                        for sk in [int(v) for v in a.splitk.split(",")]:
                            if sk > 1 and (d.KH * d.KW * d.Cin // 32) // sk < 2:
                                continue
                            if sk * d.groups * M * d.Cout > ws.numel():
                                continue
                            d.tile, d.splitk = t, sk
                            us = time_desc(lib, d, st)
                            if us is not None:
                                cands.append((us, t, sk))
                    cands.sort()
                    d.tile, d.splitk = base
                    base_us = time_desc(lib, d, st)
                    us, t, sk = cands[0]
                    table[sig] = [t, sk]
                    report.append((sig, base, base_us, (t, sk), us))
                    print("%-40s plan %-8s sk%-2d %8.1f us   best %-8s sk%-2d %8.1f us" % (
                        sig, L.TILE_NAMES[base[0]], base[1], base_us, L.TILE_NAMES[t], sk, us), flush=True)
            del sn, dc
            torch.cuda.empty_cache()
    with open(OUT, "w") as f:
        json.dump(dict(sorted(table.items())), f, indent=0)
    print("wrote %d signatures to %s" % (len(table), OUT))


if __name__ == "__main__":
    main()
