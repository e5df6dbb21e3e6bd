"""Debug: which path makes a frame's output depend on the shard (see tests/test_configs.py::test_frame_output_does_not_depend_on_the_shard)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import synthetic as S  # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask  # noqa: E402

torch.set_grad_enabled(False)
DEV = "cuda"
frames = {f: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=f).items()} for f in range(4)}
sn_sd = dc_sd = None


def make():
    global sn_sd, dc_sd
    p = DepthCompletionPipeline(enriched_samples=200, rng=np.random.RandomState(0))
    if sn_sd is None:
        sn_sd = S.seeded_state_dict(p.surface_normal_cnn.state_dict(), 1234, device=DEV)
        dc_sd = S.seeded_state_dict(p.cnn.state_dict(), 1234, device=DEV)
    p.load_state_dicts(sn_sd, dc_sd)
    p.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    return p


def run_shard(ids, taps):
    p = make()

    out = {}
    gen = p.run_interleaved(iter([frames[f] for f in ids]), frame_rng=lambda i: np.random.RandomState(1000 + ids[i]))      # (items are pulled ahead of
    for f, o in zip(ids, gen):                                                                                                #  their draws since round 4)
        out[f] = o.cpu()
    return out


for label, env in (("graph", {}), ("eager", {"VIDC_EXEC": "eager"}), ("graph, direct 3x3", {"VIDC_WINOGRAD": "0"})):
    for k in ("VIDC_EXEC", "VIDC_WINOGRAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    a = run_shard([0, 1, 2, 3], None)
    a2 = run_shard([0, 1, 2, 3], None)
    b = run_shard([1, 3], None)
    b2 = run_shard([1, 3], None)
    print(label, "A vs A again:", [bool(torch.equal(a[f], a2[f])) for f in (0, 1, 2, 3)], " B vs B again:", [bool(torch.equal(b[f], b2[f])) for f in (1, 3)],
          " A vs B:", [(bool(torch.equal(a[f], b[f])), "%.2e" % float((a[f] - b[f]).abs().max())) for f in (1, 3)])
