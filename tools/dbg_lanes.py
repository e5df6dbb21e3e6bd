import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vi_depth_completion_amd import synthetic as S, ops
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
from vi_depth_completion_amd.networks.warping_2dof_alignment import Warping2DOFAlignment
torch.set_grad_enabled(False)
DEV = "cuda"
# what tests/test_hip_parity.py::test_stem_conv_gathers_through_the_warp_bit_for_bit does first
w_ = np.load(os.path.join(ROOT, "tests", "golden", "warp_cases.npz"))
g, a = torch.from_numpy(w_["gravity"]), torch.from_numpy(w_["aligned"])
img = S.uniform01(1234, "warp.image", (1, 3, 240, 320)).repeat(g.shape[0], 1, 1, 1)
for ac in (False, True):
    wp = Warping2DOFAlignment(float(w_["fx"]), float(w_["fy"]), float(w_["cx"]), float(w_["cy"]), align_corners=ac)
    wt = S.normal01(5, "stem.w", (64, 3, 3, 3), scale=0.2).float().to(DEV)
    x = img.to(DEV)
    params = wp._params(g.to(DEV), a.to(DEV))
    _H, warped = wp.warp_with_gravity_center_aligned(x, g.to(DEV), a.to(DEV))
    ref = ops.stem_conv3x3s2(warped, wt, relu=True)
    got = ops.stem_conv3x3s2_warped(x, params, wt, wp.cx, wp.cy, ac, relu=True)
    assert torch.equal(got, ref)
del x, params, warped, ref, got, wt
pipe = DepthCompletionPipeline(enriched_samples=int(os.environ.get("DBG_ENRICH", "200")), device=torch.device(DEV), rng=np.random.RandomState(3))
sn = S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device=DEV); dc = S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device=DEV)
pipe.load_state_dicts(sn, dc)
pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
frames = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=60 + i).items()} for i in range(7)]
if os.environ.get("DBG_PREPARE", "0") == "1":
    pipe.prepare_interleaved(frames[0], lanes=3, frames_per_launch=1)
    torch.cuda.synchronize()
ref = None
for lanes in (1, 2, 3, 2, 1):
    pipe.rng = np.random.RandomState(99)
    outs = [o.cpu() for o in pipe.run_interleaved(iter(frames), lanes=lanes, copy_outputs=os.environ.get("DBG_COPY", "1") == "1")]
    if ref is None:
        ref = outs
    else:
        for f, (a_, b_) in enumerate(zip(ref, outs)):
            if not torch.equal(a_, b_):
                d = (a_ - b_).abs()
                print("lanes", lanes, "frame", f, "differs: n=%d max=%.3e nan=%d" % (int((d > 0).sum()), float(d.max()), int(torch.isnan(b_).sum())))
    print("lanes", lanes, "done")

# ---- second part: per-frame taps of the surface-normal side (DBG_TAPS=1)
if os.environ.get("DBG_TAPS", "0") == "1":
    from vi_depth_completion_amd import pipeline as P
    LOG = []
    orig_begin = P._GroupLane.begin
    def begin(self, n):
        orig_begin(self, n)
        with torch.cuda.stream(self.stream):
            prog = self.prog
            pbuf = next(kw["p"].buf for k, _r, _w, kw in prog.ops if k == "warp_params")
            allb = None
            if os.environ.get("DBG_STEM_ONLY", "0") == "1":      # just the fused stem's own slice (light: keeps the timing close to the untapped run)
                i = next(i for i, op in enumerate(prog.ops) if op[0] == "stem")
                y = prog.ops[i][3]["y"]
                allb = [(i, prog.op_names[i], [prog.storage[w][: prog.storage[w].numel() // y.ld * y.ld].view(-1, y.ld)[:, :y.C].clone() for w in prog.ops[i][2]])]
            elif os.environ.get("DBG_FEW", "0") == "1":      # a handful of group-0 slices through the net (needs VIDC_NO_BUFFER_REUSE=1)
                b0, e0 = prog.segments()[0]
                want = ("stem:sn/", "conv1.conv1_2", "conv1.conv1_3", "maxpool", "layer1.2.conv3", "layer2.3.conv3", "layer3.10.conv3", "layer3.22.conv3", "layer4.2.conv3",
                        "sn/feature1_upsamping.3", "sn/feature_concat.0", "head:sn")
                def g0(i, w):
                    kw = prog.ops[i][3]
                    y = kw.get("y")
                    t = prog.storage[w]
                    if y is not None and (y.G == 4 or prog.ops[i][0] == "stem") and not y.nchw:
                        return t[: t.numel() // y.ld * y.ld].view(-1, y.ld)[:, :y.C].clone()
                    return t.clone()
                sel = [i for i in range(b0, e0) if any(wn in prog.op_names[i] for wn in want) and not prog.op_names[i].startswith("wino")]
                allb = [(i, prog.op_names[i], [g0(i, w) for w in prog.ops[i][2]]) for i in sel]
            elif os.environ.get("VIDC_NO_BUFFER_REUSE", "0") == "1":      # every op output of segment 0, in op order
                b0, e0 = prog.segments()[0]
                def g0(i, w):      # group 0's channel slice of a grouped op output (the other groups belong to another frame)
                    kw = prog.ops[i][3]
                    y = kw.get("y")
                    t = prog.storage[w]
                    if y is not None and (y.G == 4 or prog.ops[i][0] == "stem") and not y.nchw:
                        ld, Cg = y.ld, y.C
                        return t[: t.numel() // ld * ld].view(-1, ld)[:, :Cg].clone()
                    return t.clone()
                allb = [(i, prog.op_names[i], [g0(i, w) for w in prog.ops[i][2]]) for i in range(b0, e0)]
            LOG.append((self.index, self.sn_image.clone(), prog.storage[pbuf][:32].clone(), self.normals.clone(), self.grav.clone(), allb))
    P._GroupLane.begin = begin
    runs = {}
    for lanes in (1, 2, 2, 3):
        LOG.clear()
        pipe.rng = np.random.RandomState(99)
        outs = [o.cpu() for o in pipe.run_interleaved(iter(frames), lanes=lanes)]
        torch.cuda.synchronize()
        cur = [(ln, a.cpu(), b.cpu(), c.cpu(), d.cpu(), ab) for ln, a, b, c, d, ab in LOG]
        if 1 not in runs:
            runs[1] = (cur, outs)
            continue
        r0, o0 = runs[1]
        for f in range(len(frames)):
            names = ("sn_image", "params", "normals", "gravity")
            diffs = [names[k] for k in range(4) if not torch.equal(r0[f][k + 1], cur[f][k + 1])]
            if diffs and r0[f][5] is not None:
                for (i, name, bufs0), (_i, _n, bufs1) in zip(r0[f][5], cur[f][5]):
                    bad = [k for k, (x0, x1) in enumerate(zip(bufs0, bufs1)) if not torch.equal(x0.view(torch.int32), x1.view(torch.int32))]
                    if bad:
                        x0, x1 = bufs0[bad[0]], bufs1[bad[0]]
                        ne = (x0.view(torch.int32) != x1.view(torch.int32))
                        idx = torch.nonzero(ne.reshape(-1))[:, 0]
                        print("   differing op %d %s: output %s, %d of %d words differ" % (i, name[:90], bad, int(ne.sum()), ne.numel()))
                        continue
                        if False:
                            # which input was stale?  candidates: (image, params) of this frame / of the lane's previous frame
                            L_ = lanes
                            wkey = "sn/resnet_pyramids.conv1.conv1_1.weight"
                            wt_ = pipe.surface_normal_cnn.state_dict()["resnet_pyramids.conv1.conv1_1.weight"]
                            wpa = pipe.surface_normal_cnn.warp_2dof_alignment
                            def cand(img_t, p_t):
                                yv = ops.stem_conv3x3s2_warped(img_t.to(DEV), p_t.to(DEV), wt_, wpa.cx, wpa.cy, wpa.align_corners, relu=True)
                                return ops.split_bf16x3(yv).reshape(-1, 64).cpu()
                            prev = f - L_
                            got = x1.reshape(-1, 64)
                            for label, im, pp in (("cur image, cur params", cur[f][1], cur[f][2]),
                                                  ("PREV image, cur params", cur[prev][1] if prev >= 0 else None, cur[f][2]),
                                                  ("cur image, PREV params", cur[f][1], cur[prev][2] if prev >= 0 else None),
                                                  ("PREV image, PREV params", cur[prev][1] if prev >= 0 else None, cur[prev][2] if prev >= 0 else None)):
                                if im is None or pp is None:
                                    continue
                                c = cand(im, pp).to(got.device)
                                eq = (c.view(torch.int32) == got.view(torch.int32)).float().mean().item()
                                print("      candidate %-26s: %.4f of the words equal" % (label, eq))
                        break
            if diffs or not torch.equal(o0[f], outs[f]):
                print("lanes %d frame %d (lane %d): differing taps %s; depth equal %s" % (lanes, f, cur[f][0], diffs, bool(torch.equal(o0[f], outs[f]))))
        print("taps lanes", lanes, "done")
