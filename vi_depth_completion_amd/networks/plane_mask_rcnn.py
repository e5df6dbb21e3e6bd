"""The reference's plane-mask detector network -- `GeneralizedRCNN` of plane_mask_detection/ (a maskrcnn_benchmark fork) built from
`configs/R101_bs16_all_plane_normal.yaml`: R-101-FPN backbone, RPN, 2-class box head, mask head -- with the same
648 state_dict keys in the same order, executed as three HIP engine programs:

  dense(B)   image -> [input transform + stem im2col] -> ResNet-101 (FrozenBatchNorm folded, stride in the 1x1) -> FPN -> RPN head
             modeling/backbone/resnet.py, fpn.py:50-85, rpn/rpn.py:78-110; outputs P2..P5 and one [h][w][32] RPN map per level
             (3 objectness logits + 12 box deltas: cls_logits and bbox_pred as ONE 1x1 conv)
  box(B)     pooled 7x7 features of the R proposal slots -> fc6 -> fc7 -> [cls_score | bbox_pred] (one GEMM)
             roi_heads/box_head/roi_box_feature_extractors.py:54-82, roi_box_predictors.py
  mask(B)    pooled 14x14 features of the R detection slots -> 4 x (3x3 conv + ReLU) -> ConvTranspose2d(2, stride 2) as a 1x1 conv to
             4 sub-pixel channel blocks + ReLU -> the 1x1 logits conv on the un-shuffled layout
             roi_heads/mask_head/roi_mask_feature_extractors.py:19-70, roi_mask_predictors.py:10-36

The `upconv` branch (UpConvNet) only exists as parameter containers: GeneralizedRCNN computes it at inference and discards the result
(generalized_rcnn.py:104-108, 253-254), so it is never executed here.  Everything between the programs (proposal selection, ROIAlign,
detection filtering, mask pasting, instance map) is csrc/plane_mask.hip, driven by `plane_mask.PlaneMaskDetector`.
"""
import collections

import numpy as np
import torch
import torch.nn as nn

from .. import engine
from .surface_normal import _HipModule

STAGE_BLOCKS = (3, 4, 23, 3)
PIXEL_MEAN_BGR = (102.9801, 115.9465, 122.7717)          # config/defaults.py INPUT.PIXEL_MEAN; PIXEL_STD = 1, TO_BGR255
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_STRIDES = (4, 8, 16, 32, 64)
ASPECT_RATIOS = (0.5, 1.0, 2.0)
ROI_SLOTS = 50                                            # FPN_POST_NMS_TOP_N_TEST: proposals (and so detections) per image


class FrozenBatchNorm2d(nn.Module):
    """layers/batch_norm.py: four buffers, no eps, no num_batches_tracked."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))


def _conv(cin, cout, k, bias):
    return nn.Conv2d(cin, cout, k, bias=bias)


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, project):
        super().__init__()
        if project:          # registered first, like the reference (resnet.py Bottleneck.__init__)
            self.downsample = nn.Sequential(_conv(cin, cout, 1, False), FrozenBatchNorm2d(cout))
        self.conv1, self.bn1 = _conv(cin, mid, 1, False), FrozenBatchNorm2d(mid)
        self.conv2, self.bn2 = _conv(mid, mid, 3, False), FrozenBatchNorm2d(mid)
        self.conv3, self.bn3 = _conv(mid, cout, 1, False), FrozenBatchNorm2d(cout)


def _upconv_containers():
    """Parameter layout of modeling/upconv/UpConvNet.py:30-75 for OUTPUT_REPRESENTATION 5 (indices = positions inside the reference's
    nn.Sequential, where the parameter-free Upsample / ReLU modules sit in between)."""
    def conv():
        return nn.Conv2d(256, 256, 3, 1, 1)

    def seq(layout):
        return nn.Sequential(collections.OrderedDict((str(i), m) for i, m in layout))
    up = nn.Module()
    up.conv1 = seq([(0, conv()), (1, nn.BatchNorm2d(256))])
    up.conv2 = seq([(0, conv()), (2, nn.BatchNorm2d(256))])
    up.conv3 = seq([(0, conv()), (2, nn.BatchNorm2d(256)), (4, conv()), (6, nn.BatchNorm2d(256))])
    up.conv4 = seq([(0, conv()), (2, nn.BatchNorm2d(256)), (4, conv()), (6, nn.BatchNorm2d(256)), (8, conv()), (10, nn.BatchNorm2d(256))])
    up.conv5 = seq([(0, conv()), (2, nn.BatchNorm2d(256)), (4, conv()), (6, nn.BatchNorm2d(256)), (8, conv()), (10, nn.BatchNorm2d(256)),
                    (12, conv()), (14, nn.BatchNorm2d(256))])
    up.conv_output_sttctt = seq([(0, nn.Conv2d(256, 2, 1))])
    up.conv_output_phi = seq([(0, nn.Conv2d(256, 1, 1))])
    return up


def cell_anchors(stride, size):
    """modeling/rpn/anchor_generator.py:210-289 for one FPN level: three aspect ratios of one size around a stride x stride cell."""
    def whctrs(a):
        w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        ws, hs = ws[:, None], hs[:, None]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))
    w, h, xc, yc = whctrs(np.array([0, 0, stride - 1, stride - 1], dtype=np.float64))
    ratios = np.array(ASPECT_RATIOS, dtype=np.float64)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    out = []
    for a in mk(ws, hs, xc, yc):
        w2, h2, xc2, yc2 = whctrs(a)
        sc = np.array([size], dtype=np.float64) / stride
        out.append(mk(w2 * sc, h2 * sc, xc2, yc2))
    return torch.from_numpy(np.vstack(out)).float()


def detector_arithmetic():
    """Arithmetic of the detector's three programs.  What they compute feeds DECISIONS -- score > 0.9, NMS IoU > 0.5 / 0.7, mask > 0.5 per
    pixel (demo/predictor.py:143-190, 253-323) -- which are index work and must come out as the oracle's: every conv in exact fp32, direct
    form (round 4 ran them in the mixed mode: bf16x3 products moved a box regression by ~1e-3 px and flipped one NMS decision on one of eight
    noise frames; the cost of fp32 here is ~1.3 ms per batch of 8 next to ~26 ms).  VIDC_DETECTOR_PRECISION=mixed restores the old behaviour."""
    import os
    mode = os.environ.get("VIDC_DETECTOR_PRECISION", "fp32")
    return {"mode": mode, "winograd": "0" if mode == "fp32" else None}


class GeneralizedRCNN(_HipModule):
    def __init__(self):
        super().__init__()
        body = nn.Module()
        body.stem = nn.Module()
        body.stem.conv1, body.stem.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), FrozenBatchNorm2d(64)
        cin = 64
        for li, n in enumerate(STAGE_BLOCKS):
            mid, cout = 64 << li, 256 << li
            blocks = [_Bottleneck(cin, mid, cout, True)] + [_Bottleneck(cout, mid, cout, False) for _ in range(n - 1)]
            setattr(body, "layer%d" % (li + 1), nn.Sequential(*blocks))
            cin = cout
        fpn = nn.Module()
        for lvl in range(1, 5):
            setattr(fpn, "fpn_inner%d" % lvl, nn.Conv2d(128 << lvl, 256, 1))
            setattr(fpn, "fpn_layer%d" % lvl, nn.Conv2d(256, 256, 3, 1, 1))
        self.backbone = nn.Module()
        self.backbone.body, self.backbone.fpn = body, fpn
        self.upconv = _upconv_containers()
        self.rpn = nn.Module()
        self.rpn.anchor_generator = nn.Module()
        self.rpn.anchor_generator.cell_anchors = nn.Module()
        for i, (s, z) in enumerate(zip(ANCHOR_STRIDES, ANCHOR_SIZES)):
            self.rpn.anchor_generator.cell_anchors.register_buffer(str(i), cell_anchors(s, z))
        self.rpn.head = nn.Module()
        self.rpn.head.conv = nn.Conv2d(256, 256, 3, 1, 1)
        self.rpn.head.cls_logits = nn.Conv2d(256, 3, 1)
        self.rpn.head.bbox_pred = nn.Conv2d(256, 12, 1)
        self.roi_heads = nn.Module()
        box = nn.Module()
        box.feature_extractor = nn.Module()
        box.feature_extractor.fc6, box.feature_extractor.fc7 = nn.Linear(256 * 7 * 7, 1024), nn.Linear(1024, 1024)
        box.predictor = nn.Module()
        box.predictor.cls_score, box.predictor.bbox_pred = nn.Linear(1024, 2), nn.Linear(1024, 8)
        mask = nn.Module()
        mask.feature_extractor = nn.Module()
        for i in range(1, 5):
            setattr(mask.feature_extractor, "mask_fcn%d" % i, nn.Conv2d(256, 256, 3, 1, 1))
        mask.predictor = nn.Module()
        mask.predictor.conv5_mask = nn.ConvTranspose2d(256, 256, 2, 2, 0)
        mask.predictor.mask_fcn_logits = nn.Conv2d(256, 2, 1)
        self.roi_heads.box, self.roi_heads.mask = box, mask
        for p in self.parameters():
            p.requires_grad_(False)
        self._init_engine()
        self._weights.bn_eps = 0.0                        # FrozenBatchNorm2d: weight * running_var.rsqrt()
        ws = self._weights
        # derived parameters (rebuilt with the cache whenever the state_dict changes)
        ws.add_virtual("backbone.body.stem.conv1@im2col.weight",          # columns (kh, kw, c) + 13 zero columns (vidc_det_stem_im2col)
                       lambda sd: torch.cat((sd["backbone.body.stem.conv1.weight"].permute(0, 2, 3, 1).reshape(64, 147),
                                             sd["backbone.body.stem.conv1.weight"].new_zeros(64, 13)), 1).reshape(64, 160, 1, 1).contiguous())

        def stacked(names, width):
            def f(sd, suffix):
                rows = torch.cat([sd[n + suffix].reshape(sd[n + suffix].shape[0], -1) for n in names], 0)
                pad = rows.new_zeros(32 - rows.shape[0], rows.shape[1])
                out = torch.cat((rows, pad), 0)
                return out.reshape(32, width, 1, 1).contiguous() if suffix == ".weight" else out.reshape(32).contiguous()
            return f
        rp = stacked(("rpn.head.cls_logits", "rpn.head.bbox_pred"), 256)
        ws.add_virtual("rpn.head.pred.weight", lambda sd: rp(sd, ".weight"))
        ws.add_virtual("rpn.head.pred.bias", lambda sd: rp(sd, ".bias"))
        bp = stacked(("roi_heads.box.predictor.cls_score", "roi_heads.box.predictor.bbox_pred"), 1024)
        ws.add_virtual("roi_heads.box.predictor.pred.weight", lambda sd: bp(sd, ".weight"))
        ws.add_virtual("roi_heads.box.predictor.pred.bias", lambda sd: bp(sd, ".bias"))
        ml = stacked(("roi_heads.mask.predictor.mask_fcn_logits",), 256)
        ws.add_virtual("roi_heads.mask.predictor.logits32.weight", lambda sd: ml(sd, ".weight"))
        ws.add_virtual("roi_heads.mask.predictor.logits32.bias", lambda sd: ml(sd, ".bias"))
        # ConvTranspose2d(256, 256, 2, 2): out[co, 2h+i, 2w+j] = sum_ci x[ci, h, w] W[ci, co, i, j]  ->  1x1 conv to (i, j, co)
        ws.add_virtual("roi_heads.mask.predictor.conv5_mask@1x1.weight",
                       lambda sd: sd["roi_heads.mask.predictor.conv5_mask.weight"].permute(2, 3, 1, 0).reshape(1024, 256, 1, 1).contiguous())
        ws.add_virtual("roi_heads.mask.predictor.conv5_mask@1x1.bias", lambda sd: sd["roi_heads.mask.predictor.conv5_mask.bias"].repeat(4))

    # ---- programs ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def padded(H, W, div=32):
        return (H + div - 1) // div * div, (W + div - 1) // div * div

    def build_dense(self, B, H, W, device, dry_run=False):
        Hp, Wp = self.padded(H, W)
        prog = engine.Program(self._weights, device, B, **detector_arithmetic())
        img = prog.input_nchw("image", 3, H, W)
        cols = prog.det_im2col(img, Hp, Wp, PIXEL_MEAN_BGR)
        t = prog.conv(cols, "backbone.body.stem.conv1@im2col", bn="backbone.body.stem.bn1", relu=True)
        t = prog.maxpool(t)
        feats = []
        for li, n in enumerate(STAGE_BLOCKS):
            for bi in range(n):
                p = "backbone.body.layer%d.%d." % (li + 1, bi)
                stride = 2 if (bi == 0 and li > 0) else 1
                idn = prog.conv(t, p + "downsample.0", bn=p + "downsample.1", stride=stride) if bi == 0 else t
                u = prog.conv(t, p + "conv1", bn=p + "bn1", relu=True, stride=stride)            # STRIDE_IN_1X1
                u = prog.conv(u, p + "conv2", bn=p + "bn2", relu=True, padding=1)
                t = prog.conv(u, p + "conv3", bn=p + "bn3", residual=idn, relu_after_residual=True)
            feats.append(t)
        f = "backbone.fpn."
        last = prog.conv(feats[3], f + "fpn_inner4")
        P = [None, None, None, prog.conv(last, f + "fpn_layer4", padding=1)]
        for lvl in (3, 2, 1):
            top = prog.nearest2x(last)
            last = prog.conv(feats[lvl - 1], f + "fpn_inner%d" % lvl, out=top, accumulate=True)      # lateral + top-down
            P[lvl - 1] = prog.conv(last, f + "fpn_layer%d" % lvl, padding=1)
        P.append(prog.avgpool(P[3], (1, 1), (2, 2), (0, 0)))           # LastLevelMaxPool: max_pool2d(x, 1, 2, 0) = every other pixel
        for l, p in enumerate(P):
            if l < 4:
                prog.mark_output("P%d" % (l + 2), p)
            h = prog.conv(p, "rpn.head.conv", relu=True, padding=1)
            prog.mark_output("rpn%d" % l, prog.conv(h, "rpn.head.pred"))
        prog.taps = {"C%d" % (i + 2): t_ for i, t_ in enumerate(feats)}
        prog.finalize(dry_run)
        return prog

    def build_box(self, B, device, dry_run=False):
        prog = engine.Program(self._weights, device, B * ROI_SLOTS, **detector_arithmetic())
        x = prog.nhwc(7, 7, 256)
        prog.pinned.add(x.buf)
        prog.inputs["pooled"] = x
        h = prog.linear(x, "roi_heads.box.feature_extractor.fc6", relu=True)
        h = prog.linear(h, "roi_heads.box.feature_extractor.fc7", relu=True)
        prog.mark_output("head", prog.conv(h, "roi_heads.box.predictor.pred"))
        prog.finalize(dry_run)
        return prog

    def build_mask(self, B, device, dry_run=False):
        prog = engine.Program(self._weights, device, B * ROI_SLOTS, **detector_arithmetic())
        x = prog.nhwc(14, 14, 256)
        prog.pinned.add(x.buf)
        prog.inputs["pooled"] = x
        h = x
        for i in range(1, 5):
            h = prog.conv(h, "roi_heads.mask.feature_extractor.mask_fcn%d" % i, relu=True, padding=1)
        up = prog.conv(h, "roi_heads.mask.predictor.conv5_mask@1x1", relu=True)                  # (N, 14, 14, 4 * 256)
        sub = engine.T(up.buf, up.B, 14, 56, 256)                                                # the same bytes: (N, 14, 14*4, 256)
        prog.mark_output("logits", prog.conv(sub, "roi_heads.mask.predictor.logits32"))
        prog.finalize(dry_run)
        return prog

    def programs(self, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in self._programs:
            self._programs[key] = (self.build_dense(B, H, W, device), self.build_box(B, device), self.build_mask(B, device))
        return self._programs[key]

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("GeneralizedRCNN runs through plane_mask.PlaneMaskDetector (HIP programs + detector kernels); there is no "
                           "eager/CPU path")
