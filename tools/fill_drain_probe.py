#!/usr/bin/env python3
"""Where a 20-step timed region loses time against the steady state: host time stamps of every yield of run_interleaved(lanes=2) and of
the final synchronise, for a few repetitions (the driver's bench.py run is --steps 20).
    VIDC_PRECISION=fp32 python tools/fill_drain_probe.py 20"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import synthetic as S                                      # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask    # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    H, W = 256, 320
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev, rng=np.random.RandomState(1234))
    pipe.load_state_dicts(S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device=dev), S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device=dev))
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(H, W))
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, H, W, 1234, frame0=j).items()} for j in range(4)]

    def frames(k):
        for i in range(k):
            yield pool[i % 4]

    for _ in pipe.run_interleaved(frames(8), copy_outputs=False, lanes=2):
        pass
    torch.cuda.synchronize()
    for rep in range(4):
        ts = []
        t0 = time.perf_counter()
        for _ in pipe.run_interleaved(frames(n), copy_outputs=False, lanes=2):
            ts.append(time.perf_counter() - t0)
        t_loop = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_end = time.perf_counter() - t0
        d = np.diff([0.0] + ts)
        print("rep %d: total %.3f ms (%.1f frames/s); loop returned at %.3f; yields at ms: %s" % (rep, 1e3 * t_end, n / t_end, 1e3 * t_loop, " ".join("%.2f" % (1e3 * v) for v in ts)))
        print("        steady interval (yields 6..%d): %.3f ms per frame; first yield at %.2f ms; last yield -> end %.2f ms" % (
            n - 4, 1e3 * (ts[n - 4] - ts[5]) / (n - 4 - 5), 1e3 * ts[0], 1e3 * (t_end - ts[-1])))
    # long run for the steady state
    t0 = time.perf_counter()
    for _ in pipe.run_interleaved(frames(200), copy_outputs=False, lanes=2):
        pass
    torch.cuda.synchronize()
    print("200 frames: %.3f ms per frame" % (1e3 * (time.perf_counter() - t0) / 200))


if __name__ == "__main__":
    main()
