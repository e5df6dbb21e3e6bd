"""Stress run of the stream mode for stale reads: N items, each with its own image and gravity, through
`run_interleaved(lanes, frames_per_launch)` with captured graphs, compared bit for bit with the same items through ONE lane executed
eagerly (no graphs) by a second pipeline object with the same weights.

    python tools/stale_read/stress_pipeline.py --items 2000 --lanes 3 --F 4 --runs 2
    VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1 python tools/stale_read/stress_pipeline.py ...     # the round-5 form that failed (positive control)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from vi_depth_completion_amd import synthetic as S
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask

torch.set_grad_enabled(False)
DEV = "cuda"


def make_items(n, H=240, W=320, seed=77):
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)
    images = torch.rand((n, 1, 3, H, W), device=DEV, generator=gen)
    g = torch.randn((n, 2), generator=torch.Generator().manual_seed(seed))
    grav = torch.stack([0.08 * g[:, 0], torch.ones(n), 0.12 * g[:, 1]], dim=1)
    grav = (grav / grav.norm(dim=1, keepdim=True)).float().to(DEV)
    base = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, H, W, 1234, frame0=900 + j).items()} for j in range(8)]
    items = []
    for i in range(n):
        b = dict(base[i % 8])
        b["image"] = images[i]
        b["gravity"] = grav[i:i + 1]
        items.append(b)
    return items


def make_pipe(sn=None, dc=None):
    p = DepthCompletionPipeline(enriched_samples=200, device=torch.device(DEV))
    if sn is None:
        sn = S.seeded_state_dict(p.surface_normal_cnn.state_dict(), 1234, device=DEV)
        dc = S.seeded_state_dict(p.cnn.state_dict(), 1234, device=DEV)
    p.load_state_dicts(sn, dc)
    p.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    return p, sn, dc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=2000)
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--F", type=int, default=4)
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--ref-exec", default="eager")
    a = ap.parse_args()
    os.environ.setdefault("VIDC_PRECISION", "fp32")
    items = make_items(a.items)
    rng_of = lambda i: np.random.RandomState(5000 + i)      # noqa: E731
    want_exec = os.environ.get("VIDC_EXEC", "graph")
    os.environ["VIDC_EXEC"] = a.ref_exec
    ref_pipe, sn, dc = make_pipe()
    t0 = time.time()
    ref = [o.clone() for o in ref_pipe.run_interleaved(iter(items), lanes=1, frames_per_launch=a.F, frame_rng=rng_of)]
    torch.cuda.synchronize()
    t_ref = time.time() - t0
    os.environ["VIDC_EXEC"] = want_exec
    pipe, _, _ = make_pipe(sn, dc)
    total_bad = 0
    for run in range(a.runs):
        t0 = time.time()
        got = [o.clone() for o in pipe.run_interleaved(iter(items), lanes=a.lanes, frames_per_launch=a.F, frame_rng=rng_of)]
        torch.cuda.synchronize()
        dt = time.time() - t0
        bad = [i for i, (x, y) in enumerate(zip(ref, got)) if not torch.equal(x, y)]
        total_bad += len(bad)
        detail = ", ".join("%d(n=%d max=%.1e)" % (i, int((ref[i] != got[i]).sum()), float((ref[i] - got[i]).abs().max())) for i in bad[:6])
        print("STRESS run %d: lanes=%d F=%d items=%d exec=%s fuse_warp=%s stem_loads=%s hwq=%s: %d differing items [%s] (%.1f items/s; reference %s, %.1f s)"
              % (run, a.lanes, a.F, a.items, want_exec, os.environ.get("VIDC_FUSE_WARP", "0"), os.environ.get("VIDC_DBG_STEM_LOADS", "0"),
                 os.environ.get("GPU_MAX_HW_QUEUES", "default"), len(bad), detail, a.items / dt, a.ref_exec, t_ref), flush=True)
    print("STRESS total differing items: %d" % total_bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
