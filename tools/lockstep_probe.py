#!/usr/bin/env python3
"""Two frame programs replayed side by side on two HIP streams (no host work in between): does it matter whether the two lanes share one
packed copy of the weights and whether they run in lock-step (both in the same layer at about the same time, so the trailing one finds
the weight tiles in L2 / Infinity Cache) or half a tick apart (what pipeline.run_interleaved(lanes=2) does)?

    python tools/lockstep_probe.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import engine                       # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, build_frame_program      # noqa: E402


def main():
    H, W, B = 256, 320, 1
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    table = engine.tuning_table()
    for item in [v for v in os.environ.get("VIDC_OVERRIDE", "").split(",") if v]:      # "signature:tile:splitk,..." (tile experiments)
        sig, t, sk = item.split(":")
        e = list(table[sig])
        table[sig] = [int(t), int(sk), 1] + e[3:5]
        print("override %s -> tile %s sk %s (was %s)" % (sig, t, sk, e[:2]), flush=True)

    def build(shared):
        ws = engine.JointWeightStore({"sn": pipe.surface_normal_cnn, "dc": pipe.cnn}) if shared else None
        progs = []
        for s in streams:
            with torch.cuda.stream(s):
                p = build_frame_program(pipe.surface_normal_cnn, pipe.cnn, B, H, W, dev, weights=ws)
                p.run()
                p.capture_segments()
            progs.append(p)
        torch.cuda.synchronize()
        return progs

    def run(progs, iters, stagger):
        a, b = progs
        sa, sb = streams[0].cuda_stream, streams[1].cuda_stream
        if stagger:                       # lane B starts half a tick late and stays there
            a.launch_segment(0, stream=sa)
            torch.cuda.synchronize()
        for _ in range(iters):
            if stagger:
                a.launch_segment(1, stream=sa); b.launch_segment(0, stream=sb)
                a.launch_segment(0, stream=sa); b.launch_segment(1, stream=sb)
            else:
                a.launch_segment(0, stream=sa); b.launch_segment(0, stream=sb)
                a.launch_segment(1, stream=sa); b.launch_segment(1, stream=sb)
        torch.cuda.synchronize()

    for shared in ((True,) if os.environ.get("VIDC_OVERRIDE") else (True, False)):
        progs = build(shared)
        one = min(progs[0].time(iters=20, use_graph=True, stream=streams[0].cuda_stream) for _ in range(3))
        print("one program alone: %.3f ms per tick" % one, flush=True)
        for stagger in (False, True):
            run(progs, 5, stagger)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                run(progs, 40, stagger)
                ms = 1e3 * (time.perf_counter() - t0) / 80
                best = ms if best is None else min(best, ms)
            print("weights %-8s lanes %-10s: %.3f ms per frame" % ("shared" if shared else "separate", "staggered" if stagger else "lock-step", best), flush=True)
        del progs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
