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
TILE_DIMS = {1: (128, 128), 2: (128, 64), 3: (64, 128), 4: (64, 64), 5: (64, 64), 6: (32, 64), 7: (32, 32), 8: (32, 128), 9: (32, 32),
             10: (32, 64), 11: (32, 32), 12: (32, 128), 13: (64, 64), 14: (32, 64), 15: (32, 64), 16: (32, 32), 17: (64, 64), 18: (64, 64),
             19: (64, 128), 20: (128, 64), 21: (64, 32), 22: (64, 32), 23: (64, 32), 24: (128, 128), 25: (128, 128), 26: (256, 128), 27: (128, 256),
             28: (32, 64), 29: (64, 64), 30: (32, 32), 31: (64, 128), 32: (64, 32), 33: (128, 128), 34: (128, 128), 35: (64, 64), 36: (128, 64),
             37: (64, 64), 38: (64, 32), 39: (32, 64)}


_POOL = {}


def weight_pool(nbytes, dev):
    """>= nbytes of random fp32 'weights' (grown on demand: the Winograd-domain weights of the 3072 -> 3072 layer are 1.36 GB per copy)."""
    if _POOL.get("t") is None or _POOL["t"].numel() * 4 < nbytes:
        _POOL["t"] = None
        torch.cuda.empty_cache()
        _POOL["t"] = torch.randn(max(256 << 20, (nbytes + 3) // 4), dtype=torch.float32, device=dev) * 0.05
    return _POOL["t"]


def time_desc(lib, d, st, pool, junk, copies=20):
    """GPU-bound timing with HBM-COLD weights, like in the real frame (1.5 GB of weights per frame never survive in the
    256 MB Infinity Cache until the next frame): a captured hipGraph of back-to-back launches of the op, every launch reading
    its own copy of the weights out of `pool`, and 512 MB of junk written before each timed replay.  (Launching from Python
    is host-bound at ~7 us per call and cannot rank configurations of the small layers; timing with the same weights over
    and over ranks them by their L2-hit behaviour, which is not what the frame sees.)"""
    if lib.vidc_conv2d_bn_act(C.byref(d), st) != 0:       # also sets the kernel's LDS attribute outside capture
        return None
    wbytes = d.groups * d.Cout * d.KH * d.KW * d.Cin * 4
    stride = (wbytes + 255) // 256 * 256
    pool = weight_pool(2 * stride, pool.device)          # at least two cold copies
    copies = int(max(2, min(copies, pool.numel() * 4 // stride)))
    ops = (L.Op * copies)()
    for i, o in enumerate(ops):
        o.kind = L.OP_CONV
        C.memmove(C.byref(o.u.conv), C.byref(d), C.sizeof(L.ConvDesc))
        o.u.conv.w = pool.data_ptr() + i * stride
    h = C.c_void_p()
    if lib.vidc_program_create(ops, copies, C.byref(h)) != 0:
        return None
    best = None
    try:
        if lib.vidc_program_capture(h, st) == 0:
            ms = (C.c_float * 1)()
            for rep in range(2):
                junk.fill_(rep)
                torch.cuda.synchronize()
                if lib.vidc_program_time(h, st, 1, 1, ms, None) != 0:
                    return None
                us = ms[0] * 1e3 / copies
                best = us if best is None else min(best, us)
    finally:
        lib.vidc_program_destroy(h)
    return best


def time_split(lib, x_ptr, y_ptr, rows, Cc, ldx, st, copies=20):
    """GPU-bound time of the split kernel that a bf16x3 conv needs in front of it."""
    ops = (L.Op * copies)()
    for o in ops:
        o.kind = L.OP_SPLIT
        o.u.g.p[0], o.u.g.p[1] = x_ptr, y_ptr
        o.u.g.i[0], o.u.g.i[1], o.u.g.i[2], o.u.g.i[3] = rows & 0xFFFFFFFF, rows >> 32, Cc, ldx
    h = C.c_void_p()
    if lib.vidc_program_create(ops, copies, C.byref(h)) != 0:
        return 0.0
    try:
        lib.vidc_program_run(h, st)
        if lib.vidc_program_capture(h, st) != 0:
            return 0.0
        ms = (C.c_float * 1)()
        lib.vidc_program_time(h, st, 3, 1, ms, None)
        return ms[0] * 1e3 / copies
    finally:
        lib.vidc_program_destroy(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heights", default="240,256")
    ap.add_argument("--batches", default="1")
    ap.add_argument("--splitk", default="1,2,4,8,16")
    ap.add_argument("--sigs", default="", help="comma-separated substrings: re-measure only signatures containing one of them (keeps the rest of the table)")
    ap.add_argument("--detector", action="store_true", help="measure the signatures of the plane-mask detector's three programs "
                                                             "(networks/plane_mask_rcnn.py) instead of the depth-completion path's")
    ap.add_argument("--fp32-only", action="store_true", help="re-measure only the exact-fp32 configuration of every signature (entries [3], [4] of the "
                                                              "table, and [0], [1] where the mixed mode runs the layer in fp32 too); the bf16x3 choice stays")
    ap.add_argument("--frame-only", action="store_true", help="only the signatures of the software-pipelined frame program (what bench.py times)")
    ap.add_argument("--merge", action="store_true", help="re-measure every signature of the chosen programs and merge the result over the committed table "
                                                          "(without it a plain run writes only what it measured)")
    ap.add_argument("--out", default=OUT, help="where the table is written (default: the package's conv_tuning.json)")
    ap.add_argument("--verbose", action="store_true", help="print every candidate (tile, splitk, us), fastest first, not only the winner")
    ap.add_argument("--dry", action="store_true", help="measure and print, do not write the table")
    ap.add_argument("--only-missing", action="store_true", help="keep the committed table and measure only signatures it lacks")
    ap.add_argument("--split-charge", type=float, default=0.25,
                    help="fraction of the standalone split-kernel time charged to a bf16x3 conv (most splits are fused into the\n"
                         "producing conv epilogue by engine.Program._fuse_splits, so the default charges little)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    lib = L.lib()
    engine._TUNING = {}                      # measure against the cost-model plan, not an older table
    os.environ["VIDC_PRECISION"] = "fp32"      # record the programs with fp32 inputs (no split ops); both modes are timed below
    table, report = {}, []
    if (a.only_missing or a.sigs or a.fp32_only or a.frame_only or a.merge) and os.path.exists(OUT):
        table = json.load(open(OUT))
    old_table = dict(table)
    if a.fp32_only or ((a.frame_only or a.merge) and not a.only_missing):
        table = {}                               # re-measure; everything not measured in this run is merged back before writing
    if a.sigs:
        pats = [v for v in a.sigs.split(",") if v]
        table = {k: v for k, v in table.items() if not any(pt in k for pt in pats)}
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)                 # graph capture needs a non-default stream
    st = side.cuda_stream
    ws = torch.zeros(64 << 20, dtype=torch.float32, device=dev)     # 256 MB split-K scratch (zeroed: ticket counters at its head)
    split_dst = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    pool = weight_pool(1 << 30, dev)     # 1 GB of weight copies (cold per launch); grown by time_desc when one copy is larger
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for H in [int(v) for v in a.heights.split(",")]:
        for B in [int(v) for v in a.batches.split(",")]:
            cc = np.array([0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0])
            sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0]), cc_img=cc).to(dev).eval()
            dc = ModifiedFPN().to(dev).eval()
            from vi_depth_completion_amd.pipeline import build_frame_program
            if a.detector:
                from vi_depth_completion_amd.networks.plane_mask_rcnn import GeneralizedRCNN
                det = GeneralizedRCNN().to(dev).eval()
                progs = det.programs(B, H, 320, dev)
            else:
                progs = (build_frame_program(sn, dc, B, H, 320, dev),) if a.frame_only else \
                    (sn.program(B, dev), dc.program(B, H, 320, dev), build_frame_program(sn, dc, B, H, 320, dev))
            for prog in progs:
                for op, name in zip(prog.c_ops, prog.op_names):
                    if op.kind != L.OP_CONV:
                        continue
                    sig = name.split(" ")[1]
                    if sig in table or (a.sigs and not any(pt in sig for pt in a.sigs.split(",") if pt)):
                        continue
                    d = L.ConvDesc.from_buffer_copy(op.u.conv)
                    d.workspace = ws.data_ptr()
                    d.flags &= ~L.ACCUM                     # timing launches must not accumulate into live data forever
                    M = d.B * d.Ho * d.Wo
                    best = {}
                    for prec in ((0,) if a.fp32_only else (0, 1)):
                        d.precision = prec
                        cands = []
                        for t, (bm, bn) in TILE_DIMS.items():
                            if bn > max(64, d.Cout) or bm >= 4 * max(32, M):
                                continue
                            for sk in [int(v) for v in a.splitk.split(",")]:
                                if sk > 1 and (d.KH * d.KW * d.Cin // 32) // sk < 2:
                                    continue
                                if sk > 1 and L.SPLITK_COUNTERS + sk * d.groups * M * d.Cout > ws.numel():
                                    continue
                                d.tile, d.splitk = t, sk
                                us = time_desc(lib, d, st, pool, junk)
                                if us is not None:
                                    cands.append((us, t, sk))
                        cands.sort()
                        if a.verbose:
                            print("  %s prec %d: %s" % (sig, prec, "  ".join("%s/sk%d %.1f" % (L.TILE_NAMES[t], sk, us) for us, t, sk in cands[:14])), flush=True)
                        if not cands:
                            raise RuntimeError("no tiling ran for %s (precision %d): %s" % (sig, prec, lib.vidc_last_error().decode()))
                        best[prec] = cands[0]
                    if a.fp32_only:
                        us32, t32, sk32 = best[0]
                        ent = list(old_table.get(sig, [t32, sk32, 0, t32, sk32]))
                        if len(ent) < 5:
                            ent = ent[:3] + [t32, sk32] if len(ent) == 3 else [ent[0], ent[1], 1, t32, sk32]
                        was = (L.TILE_NAMES[ent[3]], ent[4])
                        ent[3], ent[4] = t32, sk32
                        if ent[2] == 0:
                            ent[0], ent[1] = t32, sk32
                        table[sig] = ent
                        print("%-40s fp32 %-10s sk%-2d %8.1f us   (was %s sk%d)" % (sig, L.TILE_NAMES[t32], sk32, us32, was[0], was[1]), flush=True)
                        continue
                    # a bf16x3 conv needs its input split first (shared between consumers at best; charged in full here)
                    rows = d.B * d.H * d.W
                    if rows * d.Cin * d.groups > split_dst.numel():          # (the detector's mask head at batch 8: 80 M floats)
                        split_dst = torch.empty(rows * d.Cin * d.groups, dtype=torch.float32, device=dev)
                    # (the transform-domain GEMMs of a Winograd layer get their split operand from the input transform: nothing to charge)
                    t_split = 0.0 if "@wino" in name else time_split(lib, d.x, split_dst.data_ptr(), rows, d.Cin * d.groups, d.ldx, st)
                    us32, t32, sk32 = best[0]
                    us16, t16, sk16 = best[1]
                    prec = 1 if us16 + a.split_charge * t_split < 0.95 * us32 else 0
                    table[sig] = [t16, sk16, 1, t32, sk32] if prec else [t32, sk32, 0, t32, sk32]
                    print("%-40s fp32 %-8s sk%-2d %8.1f us | bf16x3 %-8s sk%-2d %8.1f us + split %5.1f us -> %s" % (
                        sig, L.TILE_NAMES[t32], sk32, us32, L.TILE_NAMES[t16], sk16, us16, t_split, "bf16x3" if prec else "fp32"), flush=True)
            del sn, dc
            torch.cuda.empty_cache()
    if a.fp32_only or a.frame_only or a.merge:
        merged = dict(old_table)
        merged.update(table)
        table = merged
    if a.dry:
        print("dry run: table not written")
        return
    with open(a.out, "w") as f:
        json.dump(dict(sorted(table.items())), f, indent=0)
    print("wrote %d signatures to %s" % (len(table), a.out))


if __name__ == "__main__":
    main()
