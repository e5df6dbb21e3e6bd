#!/usr/bin/env python3
"""How much of a small conv launch is the HBM-cold weight prologue?  For the most expensive conv signatures of the frame program (with the
committed table's tile / split-K / arithmetic), 20 back-to-back launches in one captured graph, three ways:

    cold        every launch reads its own copy of the weights and 512 MB of junk was written before the replay (what the frame sees:
                1.5 GB of weights per tick never survive in the 256 MB Infinity Cache) -- tools/autotune.py's measurement
    mall-warm   the same graph replayed again right away: the 20 copies (<= 128 MB) are still in the Infinity Cache, but not in the
                32 MB of L2 they were streamed through
    l2-warm     every launch reads the SAME weights

If mall-warm is clearly faster than cold, a prefetcher that pulls the next layers' weights into the Infinity Cache ahead of their use
would shorten the per-launch floor; if only l2-warm is, the prefetch would have to run on the consumer's XCD.

    python tools/weight_warmth_probe.py [--top 12] [--precision mixed|fp32]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import _lib as L                               # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, build_frame_program      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--precision", default="mixed")
    ap.add_argument("--copies", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="program batch (4 = what bench.py's stream mode records)")
    ap.add_argument("--sigs", default="", help="comma-separated substrings: only signatures containing one of them")
    a = ap.parse_args()
    os.environ["VIDC_PRECISION"] = a.precision
    dev = torch.device("cuda")
    lib = L.lib()
    H, W = 256, 320
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=(0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0), device=dev)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    st = side.cuda_stream
    prog = build_frame_program(pipe.surface_normal_cnn, pipe.cnn, a.batch, H, W, dev)
    prog.run()
    torch.cuda.synchronize()
    _total, per = prog.time(iters=5, use_graph=False, per_op=True)
    by_sig = {}
    for op, name, t in zip(prog.c_ops, prog.op_names, per):
        if op.kind != L.OP_CONV:
            continue
        sig = name.split(" ")[1]
        if a.sigs and not any(pt in sig for pt in a.sigs.split(",") if pt):
            continue
        e = by_sig.setdefault(sig, [0.0, 0, op, name])
        e[0] += t
        e[1] += 1
    pool = torch.randn(64 << 20, dtype=torch.float32, device=dev) * 0.05          # 256 MB of weight copies
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ms = (C.c_float * 1)()
    print("%-34s %3s %-28s %9s %10s %9s   (us per launch)" % ("signature", "n", "tile", "cold", "mall-warm", "l2-warm"))
    for sig, (t, n, op, name) in sorted(by_sig.items(), key=lambda kv: -kv[1][0])[: a.top]:
        d = L.ConvDesc.from_buffer_copy(op.u.conv)
        d.flags &= ~L.ACCUM
        wbytes = d.groups * d.Cout * d.KH * d.KW * d.Cin * 4
        stride = (wbytes + 255) // 256 * 256
        if 2 * stride > pool.numel() * 4:              # (a layer whose weights alone exceed the Infinity Cache streams them at HBM rate)
            print("%-34s %3d  weights %.1f MB: larger than the probe's pool, skipped" % (sig, n, wbytes / 1e6))
            continue
        copies = int(max(2, min(a.copies, (128 << 20) // stride)))
        res = []
        for same in (False, True):
            ops = (L.Op * copies)()
            for i, o in enumerate(ops):
                o.kind = L.OP_CONV
                C.memmove(C.byref(o.u.conv), C.byref(d), C.sizeof(L.ConvDesc))
                o.u.conv.w = pool.data_ptr() + (0 if same else i * stride)
            h = C.c_void_p()
            L.check(lib.vidc_program_create(ops, copies, C.byref(h)), "create")
            L.check(lib.vidc_program_run(h, st), "run")
            L.check(lib.vidc_program_capture(h, st), "capture")
            if not same:
                cold, warm = [], []
                for rep in range(3):
                    junk.fill_(rep)
                    torch.cuda.synchronize()
                    lib.vidc_program_time(h, st, 1, 1, ms, None)
                    cold.append(ms[0] * 1e3 / copies)
                    lib.vidc_program_time(h, st, 1, 1, ms, None)
                    warm.append(ms[0] * 1e3 / copies)
                res += [min(cold), min(warm)]
            else:
                lib.vidc_program_time(h, st, 3, 1, ms, None)
                res.append(ms[0] * 1e3 / copies)
            lib.vidc_program_destroy(h)
        print("%-34s %3d %-28s %9.2f %10.2f %9.2f   weights %.1f MB x %d copies" % (sig, n, name.split(" ")[0].split(":", 2)[2][-28:], res[0], res[1], res[2], wbytes / 1e6, copies), flush=True)


if __name__ == "__main__":
    main()
