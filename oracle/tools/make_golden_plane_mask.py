#!/usr/bin/env python3
"""Golden vectors for the plane-mask detector, produced by the REFERENCE ITSELF: the reference's GeneralizedRCNN
(imported on CPU through oracle/tools/ref_detector_shims.py) with seeded weights on a real demo frame and a synthetic frame, its
Masker, and COCODemo.select_top_predictions / overlay_mask called on a bare COCODemo object (its constructor would try to download
a checkpoint).  Also checks the restatement oracle/plane_mask_oracle.py against the reference stage by stage and prints the
differences.  Run in the build container only:  python oracle/tools/make_golden_plane_mask.py
Writes tests/golden/plane_mask_{demo,synthetic}.npz (small: probes and discrete results, no feature maps) and
tests/golden/plane_mask_manifest.npz (state_dict keys/shapes)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "tools"))
import ref_detector_shims as R                      # noqa: E402
from oracle import plane_mask_oracle as PM          # noqa: E402
from vi_depth_completion_amd import synthetic as S  # noqa: E402

torch.set_grad_enabled(False)
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    model, cfg = R.build_reference_detector()
    from plane_mask_detection.maskrcnn_benchmark.structures.image_list import to_image_list
    from plane_mask_detection.maskrcnn_benchmark.modeling.roi_heads.mask_head.inference import Masker
    from plane_mask_detection.demo.predictor import COCODemo
    ref_sd = model.state_dict()
    np.savez_compressed(os.path.join(GOLD, "plane_mask_manifest.npz"), keys=np.array(list(ref_sd.keys())),
                        shapes=np.array([str(tuple(v.shape)) for v in ref_sd.values()]),
                        anchors=np.concatenate([v.numpy() for k, v in ref_sd.items() if "anchor_generator" in k]))
    sd = S.seeded_detector_state_dict(ref_sd, 1234)
    model.load_state_dict(sd)
    demo = np.load(os.path.join(GOLD, "preprocess_demo_000000.npz"))["image"]
    images = {"demo": torch.from_numpy(demo), "synthetic": S.uniform01(1234, "plane_mask.image", (3, 240, 320))}
    for fr in ("000068", "000085"):          # the other two demo frames of the whole-path fixtures (image_u8 as DemoDataset produced it)
        u8 = np.load(os.path.join(GOLD, "demo_%s.npz" % fr))["image_u8"]
        images["demo_" + fr] = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
    coco = object.__new__(COCODemo)
    coco.confidence_threshold = 0.9
    coco.cfg = cfg
    for name, img in images.items():
        # ---- the reference: model forward with taps on its sub-modules ------------------------------------------------
        x, hw = PM.preprocess(img[None])             # transforms need torchvision/PIL round trips: restated (oracle docstring)
        il = to_image_list(x[0, :, :hw[0], :hw[1]], cfg.DATALOADER.SIZE_DIVISIBILITY)
        assert torch.equal(il.tensors, x) and il.image_sizes[0] == torch.Size(hw)
        ref = {}
        hooks = [model.backbone.register_forward_hook(lambda m, i, o: ref.__setitem__("feats", o)),
                 model.rpn.head.register_forward_hook(lambda m, i, o: ref.__setitem__("rpn", o)),
                 model.rpn.register_forward_hook(lambda m, i, o: ref.__setitem__("proposals", o[0])),
                 model.roi_heads.box.predictor.register_forward_hook(lambda m, i, o: ref.__setitem__("box_out", o)),
                 model.roi_heads.box.register_forward_hook(lambda m, i, o: ref.__setitem__("dets", o[1])),
                 model.roi_heads.mask.register_forward_hook(lambda m, i, o: ref.__setitem__("masks", o[1]))]
        result, _, _ = model(il)
        for h in hooks:
            h.remove()
        pred = result[0]
        mask_prob_ref = pred.get_field("mask").clone()
        pasted_ref = Masker(threshold=0.5, padding=1)([pred.get_field("mask")], [pred])[0]
        pred.add_field("mask", pasted_ref)
        top = COCODemo.select_top_predictions(coco, pred)
        inst_ref = COCODemo.overlay_mask(coco, np.zeros((hw[0], hw[1], 3), np.uint8), top)
        # ---- the restatement ------------------------------------------------------------------------------------------
        taps = {}
        inst = PM.run_on_tensor(sd, img, taps=taps)

        def d(a, b):
            return float((a - b).abs().max()) if a.numel() else 0.0
        print("== %s" % name)
        for l in range(5):
            print("  P%d feat max|diff| %.3e (max|ref| %.2f)   rpn logits %.3e deltas %.3e" % (
                l + 2, d(taps["feats"][l], ref["feats"][l]), float(ref["feats"][l].abs().max()),
                d(taps["rpn_logits"][l], ref["rpn"][0][l]), d(taps["rpn_deltas"][l], ref["rpn"][1][l])))
        pb = ref["proposals"][0]
        print("  proposals: %d vs %d, boxes %.3e objectness %.3e" % (len(pb), taps["proposals"].shape[0], d(pb.bbox, taps["proposals"]),
                                                                     d(pb.get_field("objectness"), taps["objectness"])))
        print("  class logits %.3e box regression %.3e" % (d(ref["box_out"][0], taps["class_logits"]), d(ref["box_out"][1], taps["box_regression"])))
        db = ref["dets"][0]
        print("  detections: %d vs %d, boxes %.3e scores %.3e labels equal %s; > 0.9: %d" % (
            len(db), taps["det_boxes"].shape[0], d(db.bbox, taps["det_boxes"]), d(db.get_field("scores"), taps["det_scores"]),
            torch.equal(db.get_field("labels"), taps["det_labels"]), int((taps["det_scores"] > 0.9).sum())))
        print("  mask prob %.3e; pasted masks equal %s; instance map equal %s, ids %s" % (
            d(mask_prob_ref, taps["mask_prob"]), torch.equal(pasted_ref[:, 0], taps["pasted"]),
            np.array_equal(inst_ref, inst), np.bincount(inst_ref.reshape(-1)).tolist()))
        obj0 = taps["rpn_logits"][0].sigmoid()
        print("  objectness P2: min %.4f max %.4f, distinct values %d of %d" % (float(obj0.min()), float(obj0.max()),
                                                                              obj0.unique().numel(), obj0.numel()))
        # ---- fixtures: everything from the REFERENCE run --------------------------------------------------------------
        probe = {}
        for l in range(5):
            f = ref["feats"][l][0]
            probe["feat%d_probe" % l] = f[::16, ::3, ::3].numpy().astype(np.float32)          # 16 channels x coarse grid
            probe["feat%d_sum" % l] = np.array([float(f.double().sum()), float(f.double().abs().sum())])
            probe["rpn_logits%d" % l] = ref["rpn"][0][l][0].numpy().astype(np.float32) if l >= 2 else ref["rpn"][0][l][0, :, ::4, ::4].numpy()
            probe["rpn_deltas%d" % l] = ref["rpn"][1][l][0].numpy().astype(np.float32) if l >= 2 else ref["rpn"][1][l][0, :, ::4, ::4].numpy()
        np.savez_compressed(os.path.join(GOLD, "plane_mask_%s.npz" % name), image=img.numpy(),
                            proposals=pb.bbox.numpy(), objectness=pb.get_field("objectness").numpy(),
                            class_logits=ref["box_out"][0].numpy(), box_regression=ref["box_out"][1].numpy(),
                            det_boxes=db.bbox.numpy(), det_scores=db.get_field("scores").numpy(), det_labels=db.get_field("labels").numpy(),
                            mask_prob=mask_prob_ref.numpy().astype(np.float16),
                            pasted_packed=np.packbits(pasted_ref[:, 0].numpy().astype(bool), axis=-1), instance_map=inst_ref, **probe)


if __name__ == "__main__":
    main()
