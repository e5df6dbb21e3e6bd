"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (imported only by tests/ and oracle/tools/).

Functional torch-CPU fp32 restatement of the reference's plane-mask detector at inference: what
`COCODemo.run_on_tensor` (plane_mask_detection/demo/predictor.py:143-150) computes for one image with the shipped config
`configs/R101_bs16_all_plane_normal.yaml` -- R-101-FPN Mask R-CNN, 2 classes -- followed by the fork's own `overlay_mask`
(instance-id map of the biggest connected component of every confident mask).  The `upconv` branch of the fork's GeneralizedRCNN is
computed by the reference but its result is discarded at inference (generalized_rcnn.py:104-108, 253-254): not restated.

Parity pin: the reference has no test for this path; this file is pinned against the reference ITSELF, imported in the build container
through oracle/tools/ref_detector_shims.py (its `_C.nms` / `_C.roi_align_forward` bound to oracle/detector_oracle.py, whose NMS is pinned
to the reference's own test vectors); oracle/tools/make_golden_plane_mask.py writes tests/golden/plane_mask_*.npz and
tests/test_plane_mask.py checks every stage of this file against them.

What each function follows (paths relative to plane_mask_detection/):
  preprocess()          demo/predictor.py:101-118 (build_transform: ToPILImage, Resize = identity at 240x320, ToTensor, BGR, x255,
                        Normalize(mean, 1)), :143-144 (uint8 cast of 255*image), structures/image_list.py to_image_list (zero pad to /32)
  backbone()            maskrcnn_benchmark/modeling/backbone/resnet.py (StemWithFixedBatchNorm, BottleneckWithFixedBatchNorm,
                        stride in the 1x1, R-101 = [3,4,23,3]), layers/batch_norm.py (FrozenBatchNorm2d: no eps)
  fpn()                 modeling/backbone/fpn.py:50-85 (top-down nearest x2 + lateral, 3x3 output convs, LastLevelMaxPool)
  rpn_head()            modeling/rpn/rpn.py:78-110 (RPNHead)
  anchors()             modeling/rpn/anchor_generator.py:44-96,210-289
  decode()              modeling/box_coder.py:62-101
  rpn_proposals()       modeling/rpn/inference.py:74-190 (per level: sigmoid, top-1000, decode, clip, NMS 0.7, 50; then top-50 per image)
  level_of(), pool()    modeling/poolers.py:11-122 (LevelMapper, per-level ROIAlign), csrc/cpu/ROIAlign_cpu.cpp
  box_head()            modeling/roi_heads/box_head/roi_box_feature_extractors.py:54-82 (FPN2MLP), roi_box_predictors.py (FPNPredictor)
  detections()          modeling/roi_heads/box_head/inference.py:47-146 (softmax, decode 10/10/5/5, clip, >0.05, NMS 0.5, <=100)
  mask_head()           modeling/roi_heads/mask_head/roi_mask_feature_extractors.py:19-70, roi_mask_predictors.py:10-36,
                        inference.py:27-49 (sigmoid, channel = label)
  paste_masks()         modeling/roi_heads/mask_head/inference.py:86-150 (Masker(0.5, padding 1))
  instance_map()        demo/predictor.py:201-220 (score > 0.9, descending), :253-323 (overlay_mask, get_biggest_plane: scipy.ndimage.label)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import detector_oracle as DO

PIXEL_MEAN = (102.9801, 115.9465, 122.7717)        # config/defaults.py INPUT.PIXEL_MEAN (BGR), PIXEL_STD = 1
SIZE_DIVISIBILITY = 32
STAGE_BLOCKS = (3, 4, 23, 3)
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_STRIDES = (4, 8, 16, 32, 64)
ASPECT_RATIOS = (0.5, 1.0, 2.0)
PRE_NMS_TOP_N, POST_NMS_TOP_N, FPN_POST_NMS_TOP_N, RPN_NMS_THRESH = 1000, 50, 50, 0.7
BBOX_XFORM_CLIP = math.log(1000.0 / 16)
SCORE_THRESH, DET_NMS_THRESH, DETECTIONS_PER_IMG = 0.05, 0.5, 100
POOLER_SCALES = (0.25, 0.125, 0.0625, 0.03125)
CONFIDENCE_THRESHOLD = 0.9                               # main.py:254


# ---- image ------------------------------------------------------------------------------------------------------------
def preprocess(image01):
    """image01 (B,3,H,W) RGB in [0,1] -> (B,3,Hp,W') BGR x255 minus mean, zero-padded to multiples of 32; also returns (H, W)."""
    B, _, H, W = image01.shape
    u8 = (255.0 * image01).to(torch.uint8)                  # np.asarray(255. * image, dtype=np.uint8): truncation
    x = u8.float().div(255)                                  # ToTensor
    x = x[:, [2, 1, 0]] * 255                                # to_bgr_transform, TO_BGR255
    x = x - torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)        # Normalize(mean, std = 1)
    Hp = (H + SIZE_DIVISIBILITY - 1) // SIZE_DIVISIBILITY * SIZE_DIVISIBILITY
    Wp = (W + SIZE_DIVISIBILITY - 1) // SIZE_DIVISIBILITY * SIZE_DIVISIBILITY
    out = torch.zeros(B, 3, Hp, Wp)
    out[:, :, :H, :W] = x
    return out, (H, W)


# ---- dense part -------------------------------------------------------------------------------------------------------
def _fbn(sd, p, x):
    scale = sd[p + ".weight"] * sd[p + ".running_var"].rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def backbone(sd, x, prefix="backbone.body."):
    x = F.relu(_fbn(sd, prefix + "stem.bn1", F.conv2d(x, sd[prefix + "stem.conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, n in enumerate(STAGE_BLOCKS):
        for bi in range(n):
            p = "%slayer%d.%d." % (prefix, li + 1, bi)
            stride = 2 if (bi == 0 and li > 0) else 1
            idn = x
            if (p + "downsample.0.weight") in sd:
                idn = _fbn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], None, stride))
            t = F.relu(_fbn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], None, stride)))          # stride in the 1x1
            t = F.relu(_fbn(sd, p + "bn2", F.conv2d(t, sd[p + "conv2.weight"], None, 1, 1)))
            t = _fbn(sd, p + "bn3", F.conv2d(t, sd[p + "conv3.weight"]))
            x = F.relu(t + idn)
        outs.append(x)
    return outs


def fpn(sd, feats, prefix="backbone.fpn."):
    def conv(name, x, pad):
        return F.conv2d(x, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"], 1, pad)

    last = conv("fpn_inner4", feats[3], 0)
    results = [conv("fpn_layer4", last, 1)]
    for lvl in (3, 2, 1):
        top = F.interpolate(last, scale_factor=2, mode="nearest")
        last = conv("fpn_inner%d" % lvl, feats[lvl - 1], 0) + top
        results.insert(0, conv("fpn_layer%d" % lvl, last, 1))
    results.append(F.max_pool2d(results[-1], 1, 2, 0))           # LastLevelMaxPool
    return results                                                # P2 .. P6


def rpn_head(sd, feats, prefix="rpn.head."):
    logits, deltas = [], []
    for f in feats:
        t = F.relu(F.conv2d(f, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], 1, 1))
        logits.append(F.conv2d(t, sd[prefix + "cls_logits.weight"], sd[prefix + "cls_logits.bias"]))
        deltas.append(F.conv2d(t, sd[prefix + "bbox_pred.weight"], sd[prefix + "bbox_pred.bias"]))
    return logits, deltas


# ---- anchors / box arithmetic -----------------------------------------------------------------------------------------
def cell_anchors(stride, size):
    """anchor_generator.py:210-289 for one level: 3 aspect ratios x 1 scale around a stride x stride cell."""
    def whctrs(a):
        w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        ws, hs = ws[:, None], hs[:, None]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))

    anchor = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    w, h, xc, yc = whctrs(anchor)
    ratios = np.array(ASPECT_RATIOS, dtype=np.float64)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    out = []
    for a in mk(ws, hs, xc, yc):
        w2, h2, xc2, yc2 = whctrs(a)
        scales = np.array([size], dtype=np.float64) / stride
        out.append(mk(w2 * scales, h2 * scales, xc2, yc2))
    return torch.from_numpy(np.vstack(out)).float()


def anchors(grid_hw, stride, size):
    """(H*W*A, 4) in (h, w, a) order."""
    gh, gw = grid_hw
    sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cell_anchors(stride, size).view(1, -1, 4)).reshape(-1, 4)


def decode(codes, boxes, weights):
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0] + 1
    heights = boxes[:, 3] - boxes[:, 1] + 1
    cx = boxes[:, 0] + 0.5 * widths
    cy = boxes[:, 1] + 0.5 * heights
    dx, dy = codes[:, 0::4] / wx, codes[:, 1::4] / wy
    dw = torch.clamp(codes[:, 2::4] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(codes[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx, pcy = dx * widths[:, None] + cx[:, None], dy * heights[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    out = torch.zeros_like(codes)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw - 1
    out[:, 3::4] = pcy + 0.5 * ph - 1
    return out


def clip(boxes, hw):
    H, W = hw
    b = boxes.clone()
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=W - 1)
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=H - 1)
    return b


def nms(boxes, scores, thr):
    """Greedy NMS, identical decisions to detector_oracle.nms (csrc/cpu/nms_cpu.cpp: IoU >= thr suppresses), vectorised per row.
    Returns kept indices in ascending index order."""
    b = boxes.numpy().astype(np.float32)
    n = b.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    order = np.argsort(-scores.numpy().astype(np.float32), kind="stable")
    suppressed = np.zeros(n, bool)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + np.float32(1))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + np.float32(1))
        inter = (w * h).astype(np.float32)
        ovr = inter / (areas[i] + areas[rest] - inter).astype(np.float32)
        suppressed[rest[ovr >= np.float32(thr)]] = True
    return torch.from_numpy(np.flatnonzero(~suppressed).astype(np.int64))


# ---- proposals --------------------------------------------------------------------------------------------------------
def rpn_proposals(logits, deltas, image_hw):
    """logits[l] (B,A,h,w), deltas[l] (B,4A,h,w) -> per image (boxes (n,4), objectness (n,)), n <= 50, descending objectness.
    Also returns the per-level records (top-k indices, decoded boxes, kept indices) for stage-wise parity checks."""
    B = logits[0].shape[0]
    per_image = [[] for _ in range(B)]
    records = []
    for lvl, (lg, dl) in enumerate(zip(logits, deltas)):
        _, A, h, w = lg.shape
        obj = lg.permute(0, 2, 3, 1).reshape(B, -1).sigmoid()
        reg = dl.view(B, A, 4, h, w).permute(0, 3, 4, 1, 2).reshape(B, -1, 4)
        k = min(PRE_NMS_TOP_N, obj.shape[1])
        top, idx = obj.topk(k, dim=1, sorted=True)
        anc = anchors((h, w), ANCHOR_STRIDES[lvl], ANCHOR_SIZES[lvl])
        for b in range(B):
            boxes = clip(decode(reg[b, idx[b]], anc[idx[b]], (1.0, 1.0, 1.0, 1.0)), image_hw)
            ws, hs = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
            ok = torch.nonzero((ws >= 0) & (hs >= 0)).squeeze(1)                 # remove_small_boxes(min_size = 0)
            boxes, sc = boxes[ok], top[b][ok]
            keep = nms(boxes, sc, RPN_NMS_THRESH)[:POST_NMS_TOP_N]
            per_image[b].append((boxes[keep], sc[keep]))
            records.append({"level": lvl, "image": b, "topk_idx": idx[b], "topk_score": top[b], "boxes": boxes, "keep": keep})
    out = []
    for b in range(B):
        boxes = torch.cat([p[0] for p in per_image[b]])
        sc = torch.cat([p[1] for p in per_image[b]])
        _, order = torch.topk(sc, min(FPN_POST_NMS_TOP_N, sc.numel()), dim=0, sorted=True)
        out.append((boxes[order], sc[order]))
    return out, records


# ---- ROI heads --------------------------------------------------------------------------------------------------------
def level_of(boxes):
    area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224 + 1e-6))
    return torch.clamp(lvl, min=2, max=5).to(torch.int64) - 2


def pool(feats, rois, resolution):
    """feats: P2..P5 (B,C,h,w); rois (K,5) = (image, x1, y1, x2, y2) -> (K,C,res,res)."""
    lv = level_of(rois[:, 1:])
    out = torch.zeros(rois.shape[0], feats[0].shape[1], resolution, resolution)
    for l in range(4):
        idx = torch.nonzero(lv == l).squeeze(1)
        if idx.numel():
            out[idx] = torch.from_numpy(DO.roi_align_forward(feats[l].numpy(), rois[idx].numpy(), POOLER_SCALES[l], resolution, resolution, 2))
    return out


def box_head(sd, feats, rois, prefix="roi_heads.box."):
    x = pool(feats, rois, 7).flatten(1)
    x = F.relu(F.linear(x, sd[prefix + "feature_extractor.fc6.weight"], sd[prefix + "feature_extractor.fc6.bias"]))
    x = F.relu(F.linear(x, sd[prefix + "feature_extractor.fc7.weight"], sd[prefix + "feature_extractor.fc7.bias"]))
    return (F.linear(x, sd[prefix + "predictor.cls_score.weight"], sd[prefix + "predictor.cls_score.bias"]),
            F.linear(x, sd[prefix + "predictor.bbox_pred.weight"], sd[prefix + "predictor.bbox_pred.bias"]))


def detections(class_logits, box_regression, proposals, image_hw):
    """One image: proposals (n,4) -> (boxes (m,4), scores (m,), labels (m,)) in the reference's order (class by class, ascending
    proposal index inside a class: the CPU NMS returns ascending indices)."""
    prob = F.softmax(class_logits, -1)
    boxes = clip(decode(box_regression, proposals, (10.0, 10.0, 5.0, 5.0)), image_hw)
    C = prob.shape[1]
    rb, rs, rl = [], [], []
    for j in range(1, C):
        inds = torch.nonzero(prob[:, j] > SCORE_THRESH).squeeze(1)
        bj, sj = boxes[inds, 4 * j:4 * j + 4], prob[inds, j]
        keep = nms(bj, sj, DET_NMS_THRESH)
        rb.append(bj[keep]); rs.append(sj[keep]); rl.append(torch.full((keep.numel(),), j, dtype=torch.int64))
    rb, rs, rl = torch.cat(rb), torch.cat(rs), torch.cat(rl)
    if rs.numel() > DETECTIONS_PER_IMG:
        th, _ = torch.kthvalue(rs, rs.numel() - DETECTIONS_PER_IMG + 1)
        keep = torch.nonzero(rs >= th.item()).squeeze(1)
        rb, rs, rl = rb[keep], rs[keep], rl[keep]
    return rb, rs, rl


def mask_head(sd, feats, rois, labels, prefix="roi_heads.mask."):
    """-> (K,1,28,28) probabilities of every detection's own class."""
    x = pool(feats, rois, 14)
    for i in range(1, 5):
        x = F.relu(F.conv2d(x, sd[prefix + "feature_extractor.mask_fcn%d.weight" % i], sd[prefix + "feature_extractor.mask_fcn%d.bias" % i], 1, 1))
    x = F.relu(F.conv_transpose2d(x, sd[prefix + "predictor.conv5_mask.weight"], sd[prefix + "predictor.conv5_mask.bias"], 2))
    logits = F.conv2d(x, sd[prefix + "predictor.mask_fcn_logits.weight"], sd[prefix + "predictor.mask_fcn_logits.bias"])
    prob = logits.sigmoid()
    return prob[torch.arange(prob.shape[0]), labels][:, None]


def paste_masks(mask_prob, boxes, image_hw, thresh=0.5, padding=1):
    """(K,1,M,M), (K,4) -> (K,H,W) uint8."""
    H, W = image_hw
    K, _, M, _ = mask_prob.shape
    out = torch.zeros(K, H, W, dtype=torch.uint8)
    scale = float(M + 2 * padding) / M
    for k in range(K):
        padded = torch.zeros(1, 1, M + 2 * padding, M + 2 * padding)
        padded[0, 0, padding:-padding, padding:-padding] = mask_prob[k, 0].float()
        b = boxes[k].float()
        w_half, h_half = (b[2] - b[0]) * .5 * scale, (b[3] - b[1]) * .5 * scale
        xc, yc = (b[2] + b[0]) * .5, (b[3] + b[1]) * .5
        box = torch.stack((xc - w_half, yc - h_half, xc + w_half, yc + h_half)).to(torch.int32)
        w, h = max(int(box[2] - box[0] + 1), 1), max(int(box[3] - box[1] + 1), 1)
        m = F.interpolate(padded, size=(h, w), mode="bilinear", align_corners=False)[0, 0] > thresh
        x0, x1 = max(int(box[0]), 0), min(int(box[2]) + 1, W)
        y0, y1 = max(int(box[1]), 0), min(int(box[3]) + 1, H)
        if x1 > x0 and y1 > y0:
            out[k, y0:y1, x0:x1] = m[y0 - int(box[1]):y1 - int(box[1]), x0 - int(box[0]):x1 - int(box[0])].to(torch.uint8)
    return out


def instance_map(masks, scores, image_hw, confidence=CONFIDENCE_THRESHOLD):
    """select_top_predictions + overlay_mask: (K,H,W) uint8 pasted masks, (K,) scores -> (H,W) uint8 ids (1 = biggest plane)."""
    from scipy import ndimage
    H, W = image_hw
    keep = torch.nonzero(scores > confidence).squeeze(1)
    keep = keep[scores[keep].sort(0, descending=True)[1]]
    sizes, biggest = [], []
    for k in keep.tolist():
        binary = (masks[k].numpy() != 0).astype(np.uint8)
        lab, nb = ndimage.label(binary)
        sz = ndimage.sum(binary, lab, range(nb + 1))
        sel = (sz == max(sz)) & (sz > 0)
        m = sel[lab]
        sizes.append(int(np.sum(m)))
        biggest.append(m)
    inst = np.zeros((H, W), np.uint8)
    index = 1
    for size, m in sorted(zip(sizes, biggest), key=lambda t: t[0], reverse=True):
        if size >= 0.05 * H * W:
            inst[m] = index
            index += 1
    return inst


# ---- whole path -------------------------------------------------------------------------------------------------------
def run_on_tensor(sd, image01_chw, confidence=CONFIDENCE_THRESHOLD, taps=None):
    """COCODemo.run_on_tensor for one (3,H,W) image in [0,1]; `taps` (dict) receives every intermediate stage."""
    x, hw = preprocess(image01_chw[None])
    feats = fpn(sd, backbone(sd, x))
    logits, deltas = rpn_head(sd, feats)
    props, recs = rpn_proposals(logits, deltas, hw)
    boxes, obj = props[0]
    rois = torch.cat((torch.zeros(boxes.shape[0], 1), boxes), 1)
    cls, reg = box_head(sd, feats[:4], rois)
    db, ds, dl = detections(cls, reg, boxes, hw)
    drois = torch.cat((torch.zeros(db.shape[0], 1), db), 1)
    mprob = mask_head(sd, feats[:4], drois, dl) if db.shape[0] else torch.zeros(0, 1, 28, 28)
    pasted = paste_masks(mprob, db, hw)
    inst = instance_map(pasted, ds, hw, confidence)
    if taps is not None:
        taps.update(input=x, feats=feats, rpn_logits=logits, rpn_deltas=deltas, rpn_records=recs, proposals=boxes, objectness=obj,
                    class_logits=cls, box_regression=reg, det_boxes=db, det_scores=ds, det_labels=dl, mask_prob=mprob, pasted=pasted)
    return inst
