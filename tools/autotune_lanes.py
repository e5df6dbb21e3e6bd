#!/usr/bin/env python3
"""Third tuning pass, for the two-lane stream mode (pipeline.run_interleaved(lanes=2), what bench.py runs): for the conv signatures that
take the most time in the frame program, try every (tile, split-K) and keep what makes TWO frame programs replayed side by side on
two HIP streams fastest.  tools/autotune.py ranks launches in isolation and tools/autotune_frame.py one tick alone; with two lanes a
tiling that leaves CUs or LDS to the other stream's kernels can win although it loses alone.  Coordinate descent, most expensive
signature first; rewrites conv_tuning.json.

    python tools/autotune_lanes.py --height 256 --top 10
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import _lib as L, engine                       # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, build_frame_program      # noqa: E402

OUT = os.path.join(ROOT, "vi_depth_completion_amd", "conv_tuning.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--budget-s", type=float, default=1500.0)
    ap.add_argument("--precision", choices=("mixed", "fp32"), default="mixed",
                    help="fp32: tune the exact-fp32 configuration of a signature (entries [3], [4] of the table; VIDC_PRECISION=fp32 programs)")
    ap.add_argument("--max-bm", type=int, default=0, help="only tiles whose BM is at most this (0: the default M-based bound)")
    ap.add_argument("--tiles", default="", help="comma-separated tile ids: only these are tried (default: all)")
    ap.add_argument("--streams", type=int, default=2, help="frame programs replayed side by side (round 4: bench.py's fp32 leg runs 3 lanes)")
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    os.environ["VIDC_PRECISION"] = a.precision
    fp32 = a.precision == "fp32"
    H, W, B = a.height, 320, a.batch
    dev = torch.device("cuda")
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev)
    table = engine.tuning_table()
    ws = engine.JointWeightStore({"sn": pipe.surface_normal_cnn, "dc": pipe.cnn})
    from vi_depth_completion_amd.pipeline import _lane_stream
    streams = [_lane_stream(dev, i) for i in range(a.streams)]      # the process-wide lane streams (one hardware queue each) the stream mode runs on
    t_start = time.perf_counter()

    def build():
        progs = []
        for s in streams:
            with torch.cuda.stream(s):
                p = build_frame_program(pipe.surface_normal_cnn, pipe.cnn, B, H, W, dev, weights=ws)
                p.run()
                p.capture_segments()
            progs.append(p)
        torch.cuda.synchronize()
        return progs

    def pair_ms(progs, iters):
        def burst(n):
            for _ in range(n):
                for k in (0, 1):
                    for p, s in zip(progs, streams):
                        p.launch_segment(k, stream=s.cuda_stream)
            torch.cuda.synchronize()
        burst(3)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            burst(iters)
            ms = 1e3 * (time.perf_counter() - t0) / (len(streams) * iters)
            best = ms if best is None else min(best, ms)
        return best

    progs = build()
    base = pair_ms(progs, a.iters)
    print("%d lanes: %.3f ms per program execution (batch %d) with the committed table" % (len(streams), base, B), flush=True)
    total, per = progs[0].time(iters=5, use_graph=False, per_op=True, stream=streams[0].cuda_stream)
    by_sig = {}
    for n, t in zip(progs[0].op_names, per):
        if n.startswith("conv:"):
            sig = n.split(" ")[1]
            by_sig[sig] = by_sig.get(sig, 0.0) + t
    order = sorted(by_sig, key=lambda k: -by_sig[k])[: a.top]
    for sig in order:
        if time.perf_counter() - t_start > a.budget_s:
            print("time budget used up; stopping", flush=True)
            break
        ent = list(table[sig])
        if fp32 and len(ent) < 5:
            ent = [ent[0], ent[1], ent[2] if len(ent) > 2 else 0, ent[0], ent[1]]
        cur = (ent[3], ent[4]) if fp32 else (ent[0], ent[1])
        best = (base, cur[0], cur[1])

        def entry(t, sk):
            if not fp32:
                return [t, sk, 1, ent[3], ent[4]] if len(ent) >= 5 else [t, sk, 1]
            return ([t, sk, 0, t, sk] if ent[2] == 0 else [ent[0], ent[1], ent[2], t, sk])
        M = int(sig.split("_")[0][1:])
        K = int(sig.split("_")[2][1:])
        sks = sorted({1, max(1, cur[1] - 1), cur[1], cur[1] + 1, 2 * cur[1]})
        only = [int(v) for v in a.tiles.split(",") if v]
        for t in (only if only else range(1, L.TILE_COUNT)):
            bm = int(L.TILE_NAMES[t].split("x")[0])
            if bm >= 4 * max(32, M) or (a.max_bm and bm > a.max_bm):
                continue
            for sk in sks:
                if sk > 1 and (K // 32) // sk < 2:
                    continue
                if (t, sk) == cur:
                    continue
                table[sig] = entry(t, sk)
                try:
                    ps = build()
                    ms = pair_ms(ps, 12)
                except RuntimeError:
                    continue
                if ms < best[0] * 0.995:           # 0.5 % hysteresis against timing noise
                    ms2 = pair_ms(ps, a.iters)
                    if ms2 < best[0] * 0.995:
                        best = (ms2, t, sk)
                del ps
        table[sig] = entry(best[1], best[2])
        print("%-34s %6.1f us/tick: %-10s sk%-2d -> %-10s sk%-2d   %.3f -> %.3f ms per frame" % (sig, by_sig[sig] * 1e3, L.TILE_NAMES[cur[0]], cur[1],
              L.TILE_NAMES[best[1]], best[2], base, best[0]), flush=True)
        base = best[0]
    with open(a.out, "w") as f:
        json.dump(dict(sorted(table.items())), f, indent=0)
    print("wrote %s; %.3f ms per frame" % (a.out, base))


if __name__ == "__main__":
    main()
