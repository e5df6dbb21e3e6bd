#!/usr/bin/env python3
"""Timeline of a short `run_interleaved(lanes=L, frames_per_launch=F)` stream (round 4): device time stamps (HIP events on the lanes'
streams) around every segment 0 / segment 1 / drain and host time stamps of every step of the draw sequence, relative to the start
of the stream.  Shows where a 20-step region loses time against the steady state (fill, drain, bubbles of a lane between its
segments).
    VIDC_PRECISION=fp32 python tools/group_timeline.py 20 2 2        # frames, lanes, frames per launch"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import pipeline as P, synthetic as S                       # noqa: E402

LOG = []


def _ev(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e


def _wrap(cls, name, label):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        h0 = time.perf_counter()
        e0 = _ev(self.stream)
        r = orig(self, *a, **k)
        e1 = _ev(self.stream)
        LOG.append((label, self.index, a[0] if a and isinstance(a[0], int) else None, h0, time.perf_counter(), e0, e1))
        return r
    setattr(cls, name, f)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    H, W = 256, 320
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = P.DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev, rng=np.random.RandomState(1234))
    pipe.load_state_dicts(S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device=dev), S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device=dev))
    pipe.plane_masks_extraction = P.FixedPlaneMask(S.plane_id_map(H, W))
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, H, W, 1234, frame0=j).items()} for j in range(4)]

    def frames(k):
        for i in range(k):
            yield pool[i % 4]

    for _ in pipe.run_interleaved(frames(3 * lanes * F), copy_outputs=False, lanes=lanes, frames_per_launch=F):
        pass
    torch.cuda.synchronize()
    for name, label in (("begin", "seg0"), ("decoder", "seg1"), ("drain", "drain"), ("hypotheses", "hyp"), ("enrich", "enr"), ("put", "put")):
        _wrap(P._GroupLane, name, label)
    for rep in range(2):
        LOG.clear()
        torch.cuda.synchronize()
        main_stream = torch.cuda.current_stream()
        e_start = torch.cuda.Event(enable_timing=True)
        e_start.record(main_stream)
        for ln in range(lanes):          # the lanes' streams start from the same instant
            pipe.__dict__.get("_group_lane_cache", {}).get((ln, F), {}).get("stream", main_stream).wait_event(e_start)
        t0 = time.perf_counter()
        yields = []
        for _ in pipe.run_interleaved(frames(n), copy_outputs=False, lanes=lanes, frames_per_launch=F):
            yields.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print("=== rep %d: %d frames, %d lanes, %d per launch: %.3f ms = %.1f frames/s" % (rep, n, lanes, F, 1e3 * total, n / total))
        print("    yields at ms: " + " ".join("%.2f" % (1e3 * v) for v in yields))
        busy = {}
        for label, lane, j, h0, h1, e0, e1 in LOG:
            d0, d1 = e_start.elapsed_time(e0), e_start.elapsed_time(e1)
            print("    %-5s lane %d %-4s host %7.3f .. %7.3f   device %7.3f .. %7.3f (%6.3f ms)" % (label, lane, "" if j is None else "j=%d" % j, 1e3 * (h0 - t0), 1e3 * (h1 - t0), d0, d1, d1 - d0))
            if label in ("seg0", "seg1", "drain"):
                busy.setdefault(lane, []).append((d0, d1))
        for lane, iv in sorted(busy.items()):
            iv.sort()
            gaps = ["%.2f" % (b[0] - a[1]) for a, b in zip(iv, iv[1:])]
            print("    lane %d: first segment starts %.2f, last ends %.2f, segment time %.2f ms, gaps between segments: %s" % (
                lane, iv[0][0], iv[-1][1], sum(b - a for a, b in iv), " ".join(gaps)))


if __name__ == "__main__":
    main()
