"""Debug: tests/test_frames_per_launch.py::test_item_bits_do_not_depend_on_partner_slot_lanes_or_tail outside pytest, with the size of every
difference (environment switches are read by the package as usual).  usage: python tools/dbg_fpl.py [F] [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
os.environ["VIDC_PRECISION"] = mode
import numpy as np, torch
from vi_depth_completion_amd import synthetic as S
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
torch.set_grad_enabled(False)
DEV = "cuda"
pipe = DepthCompletionPipeline(enriched_samples=200)
sn = S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device=DEV); dc = S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device=DEV)
pipe.load_state_dicts(sn, dc)
pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
frames = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=300 + i).items()} for i in range(7)]
rng_of = lambda f: np.random.RandomState(9000 + f)


def run(first, last, lanes):
    return [o.cpu() for o in pipe.run_interleaved(iter(frames[first:last]), lanes=lanes, frames_per_launch=F, frame_rng=lambda i: rng_of(first + i))]


ref = run(0, 7, 1)
bad = 0
for rep in range(int(os.environ.get("DBG_REPS", "3"))):
    for lanes in (1, 2, 3):
        got = run(0, 7, lanes)
        for f, (a, b) in enumerate(zip(ref, got)):
            if not torch.equal(a, b):
                d = (a - b).abs()
                bad += 1
                print("rep %d lanes %d frame %d differs: n=%d max=%.3e" % (rep, lanes, f, int((d > 0).sum()), float(d.max())), flush=True)
print("F=%d %s: %d mismatching frames" % (F, mode, bad))
