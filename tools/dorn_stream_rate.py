#!/usr/bin/env python3
"""Frames/s of the `use_gravity=False` path (SurfaceNormalDORN, main.py:244-245): back-to-back `_call_cnn` against
`run_interleaved` (two programs on two HIP streams, `pipeline._run_interleaved_two_programs`).
    python tools/dorn_stream_rate.py [frames]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    pipe = DepthCompletionPipeline(enriched_samples=200, device=dev, rng=np.random.RandomState(1), use_gravity=False)
    pipe.cnn.load_state_dict(S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device=dev))
    pipe.surface_normal_cnn.load_state_dict(S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device=dev))
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=j).items()} for j in range(4)]
    for mode in ("sequential", "interleaved", "sequential", "interleaved"):
        def run(k):
            if mode == "sequential":
                for i in range(k):
                    pipe._call_cnn(pool[i % 4])
            else:
                for _ in pipe.run_interleaved((pool[i % 4] for i in range(k)), copy_outputs=False):
                    pass
            torch.cuda.synchronize()
        run(10)
        t0 = time.perf_counter()
        run(n)
        dt = time.perf_counter() - t0
        print("%-12s %d frames: %.3f ms per frame (%.1f frames/s)" % (mode, n, 1e3 * dt / n, n / dt))


if __name__ == "__main__":
    main()
