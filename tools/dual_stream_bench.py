#!/usr/bin/env python3
"""Experiment: TWO software-pipelined frame streams (pipeline.run_interleaved) on two HIP streams of one GPU, even / odd frames, so that
the fixed cost of one stream's launches (dispatch, prologue, split-K epilogue, drain: ~9 of the ~14 us of a small-layer launch) can
hide under the other stream's main loops.  Workgroups of the default tilings reserve 96-128 KB of LDS, so two kernels rarely share a
CU; VIDC_LDS_CAP_KB=80 swaps in <= 80 KB tilings.

    python tools/dual_stream_bench.py --frames 200 [--streams 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import synthetic as S  # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--height", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    H, W = a.height, 320
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, H, W, 1234, frame0=i).items()} for i in range(4)]
    pipes, streams = [], []
    for i in range(a.streams):
        cc = (0.5 * 319.87654 * W / 320.0, 0.5 * 239.87603 * H / 240.0)
        p = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev, rng=np.random.RandomState(0))
        if i == 0:
            sn_sd = S.seeded_state_dict(p.surface_normal_cnn.state_dict(), 1234, device=dev)
            dc_sd = S.seeded_state_dict(p.cnn.state_dict(), 1234, device=dev)
        p.load_state_dicts(sn_sd, dc_sd)
        p.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(H, W))
        pipes.append(p)
        streams.append(torch.cuda.Stream())

    def run(n):
        def feed(k):
            for f in range(k, n, a.streams):
                yield pool[f % len(pool)]
        gens = [p.run_interleaved(feed(k), copy_outputs=False) for k, p in enumerate(pipes)]
        done, alive = 0, [True] * a.streams
        while any(alive):
            for k, g in enumerate(gens):
                if not alive[k]:
                    continue
                with torch.cuda.stream(streams[k]):
                    try:
                        next(g)
                        done += 1
                    except StopIteration:
                        alive[k] = False
        torch.cuda.synchronize()
        return done

    run(a.warmup)
    t0 = time.perf_counter()
    n = run(a.frames)
    dt = time.perf_counter() - t0
    print(json.dumps({"streams": a.streams, "frames": n, "fps": round(n / dt, 1), "ms_per_frame": round(1e3 * dt / n, 3),
                      "lds_cap_kb": os.environ.get("VIDC_LDS_CAP_KB")}))


if __name__ == "__main__":
    main()
