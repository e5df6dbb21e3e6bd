"""Where do the wrong words of the fused stem (round-5 form, VIDC_DBG_STEM_LOADS=3) come from?  Taps the stem's own output slice, the
image and the parameter records behind every segment 0 of a multi-lane stream, recomputes the stem on the quiescent device from the tapped
inputs and classifies every wrong output row: which pixels, and does the row equal the stem of the lane's PREVIOUS image (stale read) or of
its NEXT one (overwritten early)?

    VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 python tools/stale_read/diag_stem.py --items 240
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from vi_depth_completion_amd import ops
from vi_depth_completion_amd import pipeline as P
from stress_pipeline import make_items, make_pipe

torch.set_grad_enabled(False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=240)
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--F", type=int, default=4)
    a = ap.parse_args()
    items = make_items(a.items)
    pipe, _, _ = make_pipe()
    LOG, order = [], {"n": 0, "cur": {}}
    orig_put, orig_begin = P._GroupLane.put, P._GroupLane.begin

    def put(self, j, batch):
        order["cur"].setdefault(self.index, []).append(order["n"])
        order["n"] += 1
        return orig_put(self, j, batch)

    def begin(self, n):
        orig_begin(self, n)
        idx = order["cur"].pop(self.index)
        with torch.cuda.stream(self.stream):
            prog = self.prog
            i = next(i for i, op in enumerate(prog.ops) if op[0] == "stem")
            kw = prog.ops[i][3]
            y = kw["y"]
            outs = []
            for w in prog.ops[i][2]:
                t = prog.storage[w]
                outs.append(t[: t.numel() // y.ld * y.ld].view(-1, y.ld)[:, :y.C].clone())
            pbuf = next(k["p"].buf for kind, _r, _w, k in prog.ops if kind == "warp_params")
            LOG.append((self.index, idx, self.sn_image.clone(), prog.storage[pbuf][: 32 * self.sn_image.shape[0]].clone(), outs, prog.op_names[i]))

    P._GroupLane.put, P._GroupLane.begin = put, begin
    rng_of = lambda i: np.random.RandomState(5000 + i)      # noqa: E731
    for _ in pipe.run_interleaved(iter(items), lanes=a.lanes, frames_per_launch=a.F, frame_rng=rng_of):
        pass
    torch.cuda.synchronize()
    P._GroupLane.put, P._GroupLane.begin = orig_put, orig_begin
    wt = pipe.surface_normal_cnn.state_dict()["resnet_pyramids.conv1.conv1_1.weight"]
    wpa = pipe.surface_normal_cnn.warp_2dof_alignment
    mixed = os.environ.get("VIDC_PRECISION", "mixed") == "mixed"

    def stem_of(img, par):
        yv = ops.stem_conv3x3s2_warped(img, par, wt, wpa.cx, wpa.cy, wpa.align_corners, relu=True)
        return (ops.split_bf16x3(yv) if mixed else yv).reshape(-1, 64).view(torch.int32)

    print("ticks tapped: %d, stem op: %s, outputs per tick: %d" % (len(LOG), LOG[0][5], len(LOG[0][4])))
    by_lane = {}
    n_bad_ticks = 0
    for t, (lane, idx, img, par, outs, _name) in enumerate(LOG):
        FB = img.shape[0]
        n = len(idx)
        # the tapped inputs must be the items themselves
        for j, it in enumerate(idx):
            if not torch.equal(img[j], items[it]["image"][0]):
                print("tick %d lane %d slot %d: the lane's image buffer is NOT item %d's image" % (t, lane, j, it))
        got = outs[-1].reshape(-1, 64).view(torch.int32)      # (mixed: the split image is the last write; fp32: the only one)
        want = stem_of(img, par)
        rows_per = got.shape[0] // FB
        prev = by_lane.get(lane)
        by_lane[lane] = (img, par, idx)
        for j in range(n):
            g, w = got[j * rows_per:(j + 1) * rows_per], want[j * rows_per:(j + 1) * rows_per]
            ne = (g != w)
            if not bool(ne.any()):
                continue
            n_bad_ticks += 1
            rows = torch.nonzero(ne.any(dim=1))[:, 0]
            oy, ox = rows // 160, rows % 160
            msg = "tick %d lane %d slot %d item %d: %d words in %d output pixels differ; rows oy %d..%d, ox %d..%d" % (
                t, lane, j, idx[j], int(ne.sum()), rows.numel(), int(oy.min()), int(oy.max()), int(ox.min()), int(ox.max()))
            # hypotheses
            hyp = []
            if prev is not None:
                pimg, ppar, _pidx = prev
                mix_img = img.clone(); mix_img[j] = pimg[j]
                mix_par = par.clone(); mix_par[32 * j:32 * j + 32] = ppar[32 * j:32 * j + 32]
                for label, im, pp in (("PREV image, cur params", mix_img, par), ("cur image, PREV params", img, mix_par), ("PREV image, PREV params", mix_img, mix_par)):
                    c = stem_of(im, pp)[j * rows_per:(j + 1) * rows_per]
                    hyp.append("%s: %d/%d wrong rows explained" % (label, int((c[rows] == g[rows]).all(dim=1).sum()), rows.numel()))
            zero = int((g[rows] == 0).all(dim=1).sum())
            hyp.append("all-zero rows: %d" % zero)
            print(msg + " | " + "; ".join(hyp))
            if n_bad_ticks <= 3:
                print("    pixels (oy,ox):", [(int(a_), int(b_)) for a_, b_ in zip(oy[:24], ox[:24])])
    print("DIAG: %d (tick, slot) pairs with a wrong stem output out of %d items" % (n_bad_ticks, a.items))


if __name__ == "__main__":
    main()
