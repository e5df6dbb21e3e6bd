#!/usr/bin/env python3
"""Builds the software-pipelined frame program (320x256; batch = VIDC_REPLAY_BATCH, default 4 = what run_interleaved(frames_per_launch=4)
records for a batch-1 stream since round 4) and replays its two hipGraph segments N times: the smallest process
that runs the frame's kernels in their real order with HBM-cold weights -- a target for `rocprofv3 --pmc` passes (counter collection on
the whole bench.py process crashes inside rocprofv3 on this pool).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f --output-format csv -- python3 tools/frame_replay.py 30
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, build_frame_program      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    H, W, B = 256, 320, int(os.environ.get("VIDC_REPLAY_BATCH", "4"))
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        prog = build_frame_program(pipe.surface_normal_cnn, pipe.cnn, B, H, W, dev)
        prog.run()
        if os.environ.get("VIDC_EXEC", "graph") == "graph":
            prog.capture_segments()
        torch.cuda.synchronize()
        for _ in range(n):
            for k in (0, 1):
                prog.launch_segment(k) if prog.captured else prog.run_segment(k)
        torch.cuda.synchronize()
    print("replayed %d ticks" % n)


if __name__ == "__main__":
    main()
