#!/usr/bin/env python3
"""Where the HOST's time goes per frame in the two-lane stream mode (the mode bench.py times): cProfile over N frames of
pipeline.run_interleaved on device-resident synthetic frames, plus the wall time per frame and the time the host spends blocked in the
one wait per frame (PlaneBlock.enrich -> event.synchronize).
    python tools/host_profile.py [frames] [lanes] [batch] [plane-head: 0|1] [items per launch]     (batch > 1: 320x240 frames, BASELINE configs[2] / [3] shapes)"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import synthetic as S                                      # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask    # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    head = len(sys.argv) > 4 and sys.argv[4] == "1"
    F = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    H, W = (256 if B == 1 else 240), 320
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    import bench
    pipe, _sn, _dc, _cc, _det = bench.build_pipeline(H, W, dev, head)
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(B, H, W, 1234, frame0=j * B).items()} for j in range(4)]

    def run(k):
        for _ in pipe.run_interleaved((pool[i % 4] for i in range(k)), copy_outputs=False, lanes=lanes, frames_per_launch=F):
            pass
        torch.cuda.synchronize()

    run(20)
    t0 = time.perf_counter()
    run(n)
    wall = time.perf_counter() - t0
    print("%d batches of %d, %d lanes, plane head %s: %.3f ms per batch wall (%.1f frames/s)" % (n, B, lanes, head, 1e3 * wall / n, n * B / wall))
    pr = cProfile.Profile()
    pr.enable()
    run(n)
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    print(s.getvalue()[:5000])


if __name__ == "__main__":
    main()
