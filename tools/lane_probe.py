"""Frames/s of pipeline.run_interleaved at 320x256, batch 1, with (enriched_samples=200) and without (0) the plane block / enrichment,
for one and two lanes: the probe that showed the host's wait for the enrichment counts -- not the GPU -- to bound the two-lane mode
(489 vs 617 frames/s) before that wait was deferred.

    python tools/lane_probe.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vi_depth_completion_amd import synthetic as S  # noqa: E402
torch.set_grad_enabled(False)
dev = torch.device("cuda")
H, W = 256, 320
r = bench.build_pipeline(H, W, dev)
pipe = r[0] if isinstance(r, tuple) else r
pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, H, W, 1234, frame0=i).items()} for i in range(4)]
def frames(n):
    for i in range(n):
        yield pool[i % 4]
host_pool = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.items()} for b in pool]
for es, name, src in ((200, "device-resident frames", pool), (200, "HOST-resident frames (pinned staging + async copy inside the timed region)", host_pool), (0, "device-resident, no plane block", pool)):
    def frames(n, src=src):
        for i in range(n):
            yield src[i % 4]
    print(name, flush=True)
    pipe.args.enriched_samples = es
    for lanes in (1, 2):
        for out in pipe.run_interleaved(frames(30), copy_outputs=False, lanes=lanes): pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for out in pipe.run_interleaved(frames(400), copy_outputs=False, lanes=lanes): pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("enriched_samples=%d lanes=%d: %.1f fps (%.3f ms)" % (es, lanes, 400 / dt, 1e3 * dt / 400), flush=True)
