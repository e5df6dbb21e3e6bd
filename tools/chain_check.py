"""Debug/validation tool for the persistent chain kernel on the WHOLE frame program (GPU): (1) run-to-run determinism, (2) chain vs
separate launches on the same tiling (bit-identical expected), (3) independence of a group's result from the other groups' inputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import synthetic as S  # noqa: E402
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN  # noqa: E402
from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction  # noqa: E402
from vi_depth_completion_amd.pipeline import build_frame_program  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda")
H, W = 240, 320
sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).to(dev).eval()
dc = ModifiedFPN().to(dev).eval()
sn.load_state_dict(S.seeded_state_dict(sn.state_dict(), 1234, device=dev))
dc.load_state_dict(S.seeded_state_dict(dc.state_dict(), 1234, device=dev))


def build(chain, force):
    os.environ["VIDC_CHAIN"] = "1" if chain else "0"
    if force:
        os.environ["VIDC_FORCE_TILE"] = "13"
    else:
        os.environ.pop("VIDC_FORCE_TILE", None)
    return build_frame_program(sn, dc, 1, H, W, dev)


def feed(p, f_sn, f_dc, scale=1.0):
    a, b = S.synthetic_batch(1, H, W, 1234, frame0=f_sn), S.synthetic_batch(1, H, W, 1234, frame0=f_dc)
    p.tensor(p.inputs["sn_image"]).copy_(a["image"].to(dev))
    p.storage[p.inputs["gravity"].buf][:3].copy_(a["gravity"].to(dev).reshape(-1))
    p.storage[p.inputs["aligned"].buf][:3].copy_(a["aligned_direction"].to(dev).reshape(-1))
    p.tensor(p.inputs["dc_image"]).copy_(b["image"].to(dev) * scale)
    p.tensor(p.inputs["dc_normal"]).copy_(torch.nn.functional.normalize(b["image"].to(dev) - 0.5, dim=1))
    p.tensor(p.inputs["dc_depth"]).copy_(b["sparse_depth"].to(dev) * scale)


def outs(p):
    torch.cuda.synchronize()
    return p.tensor(p.outputs["normals"]).clone(), p.tensor(p.outputs["depth"]).clone()


pc = build(True, False)
print("ops:", len(pc.op_names), "chains:", [n[:60] for n in pc.op_names if n.startswith("chain")])
# (1) determinism
feed(pc, 1, 0)
pc.run(); pc.check_chains()
n0, d0 = outs(pc)
bad = 0
for it in range(20):
    feed(pc, 1, 0)
    pc.run()
    n1, d1 = outs(pc)
    bad += int(not (torch.equal(n0, n1) and torch.equal(d0, d1)))
print("(1) non-deterministic repeats:", bad, "of 20")
# (3) independence of group 0 (surface normal) from the depth-completion groups' inputs
feed(pc, 1, 5, scale=3.0)
pc.run()
n2, d2 = outs(pc)
print("(3) normals identical when the DC inputs change:", bool(torch.equal(n0, n2)), "max diff %.3e" % float((n0 - n2).abs().max()))
feed(pc, 7, 0)
pc.run()
n3, d3 = outs(pc)
print("(3) depth identical when the SN input changes:", bool(torch.equal(d0, d3)), "max diff %.3e" % float((d0 - d3).abs().max()))
# (2) chain vs separate launches on the same tiling
pa, pb = build(True, True), build(False, True)
for f in range(3):
    feed(pa, f + 1, f); feed(pb, f + 1, f)
    pa.run(); pb.run(); pa.check_chains()
    na, da = outs(pa); nb, db = outs(pb)
    print("(2) frame %d: normals equal %s (%.3e), depth equal %s (%.3e)" % (f, bool(torch.equal(na, nb)), float((na - nb).abs().max()),
                                                                          bool(torch.equal(da, db)), float((da - db).abs().max())))

# (4) graph replay of the two segments vs eager execution, different inputs on consecutive replays
pg = build(True, False)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    feed(pg, 1, 0)
    pg.run(); pg.check_chains()
    pg.capture_segments()
torch.cuda.synchronize()
for f in range(3):
    feed(pc, f + 2, f + 1); feed(pg, f + 2, f + 1)
    pc.run()
    pg.launch_segment(0); pg.launch_segment(1)
    ne, de = outs(pc); ng, dg = outs(pg)
    pg.check_chains()
    print("(4) replay %d: normals equal %s (%.3e), depth equal %s (%.3e)" % (f, bool(torch.equal(ne, ng)), float((ne - ng).abs().max()),
                                                                           bool(torch.equal(de, dg)), float((de - dg).abs().max())))
# (5) time of the chain op alone: eager, events around it
ci = next(i for i, n in enumerate(pc.op_names) if n.startswith("chain"))
import ctypes as C
from vi_depth_completion_amd import _lib as L
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize()
    e0.record()
    L.check(L.lib().vidc_program_run_range(pc.handle, L.current_stream(), ci, ci + 1), "run_range")
    e1.record()
    torch.cuda.synchronize()
    print("(5) chain op alone, eager: %.1f us" % (1e3 * e0.elapsed_time(e1)))
