"""Drop-in for the reference's plane-mask predictor `COCODemo` (plane_mask_detection/demo/predictor.py:13-150 as used by
main.py:254, 273: `COCODemo(cfg, min_image_size=240, confidence_threshold=0.9).run_on_tensor(image)`), SURVEY.md §8f-1: the instance-id
map of the confident planes of one image, computed entirely on the GPU -- three engine programs (networks/plane_mask_rcnn.py) and the
detector kernels of csrc/plane_mask.hip / csrc/detector.hip, static shapes, no host synchronisation before the result is read.

    det = PlaneMaskDetector(device="cuda"); det.load_state_dict(checkpoint["model"])
    ids = det.run_on_tensor(image_chw_01)            # (H, W) uint8 numpy, like the reference
    ids = det.run_on_batch(images_b3hw_01)           # (B, H, W) uint8 device tensor (what the pipeline consumes)

`nms_inclusive`: the reference's CPU NMS suppresses at IoU >= threshold (csrc/cpu/nms_cpu.cpp:60), its CUDA NMS at IoU > threshold
(csrc/cuda/nms.cu:57).  Default = the CPU rule, which is what the oracle (the reference imported on CPU) uses.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from .networks import plane_mask_rcnn as N

PRE_NMS_TOP_N, POST_NMS_TOP_N, RPN_NMS_THRESH = 1000, 50, 0.7          # configs/R101_bs16_all_plane_normal.yaml, config/defaults.py
SCORE_THRESH, DET_NMS_THRESH = 0.05, 0.5
MASK_SIZE, MASK_THRESH, MIN_PLANE_FRACTION = 28, 0.5, 0.05


class _Buffers:
    """Everything between the programs for one (B, H, W): allocated once, reused every frame."""

    def __init__(self, B, H, W, Hp, Wp, device):
        R = N.ROI_SLOTS
        self.level_hw = [(Hp // s, Wp // s) for s in N.ANCHOR_STRIDES[:4]]          # P2..P5: Hp, Wp are multiples of 32
        self.level_hw.append(((self.level_hw[3][0] - 1) // 2 + 1, (self.level_hw[3][1] - 1) // 2 + 1))     # P6 = max_pool2d(P5, 1, 2, 0)
        self.level_k = [min(PRE_NMS_TOP_N, 3 * h * w) for h, w in self.level_hw]
        self.level_off = np.concatenate(([0], np.cumsum(self.level_k))).astype(np.int32)
        slots = int(self.level_off[-1])
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.slots = slots
        self.boxes = torch.zeros(B, slots, 4, **f32)
        self.scores = torch.zeros(B, slots, **f32)
        self.keep = torch.zeros(B, slots, **i32)
        self.n_keep = torch.zeros(B, len(self.level_hw), **i32)
        nl = len(self.level_hw)
        self.anchors_arr = np.stack([N.cell_anchors(s, z).numpy() for s, z in zip(N.ANCHOR_STRIDES, N.ANCHOR_SIZES)]).astype(np.float32).copy()
        self.level_hw_arr = np.asarray(self.level_hw, dtype=np.int32).copy()
        self.strides_arr = np.asarray(N.ANCHOR_STRIDES, dtype=np.int32).copy()
        # NMS segments: one per (image, level) for the RPN, one per image for the detections (vidc_nms_segmented)
        self.rpn_seg_off = torch.tensor([b * slots + int(self.level_off[l]) for b in range(B) for l in range(nl)], **i32)
        self.rpn_seg_n = torch.tensor([self.level_k[l] for b in range(B) for l in range(nl)], **i32)
        self.det_seg_off = torch.arange(B, **i32) * R
        self.det_seg_n = torch.full((B,), R, **i32)
        self.nms_scratch = torch.empty(max(L.lib().vidc_nms_segmented_scratch_bytes(B * nl, max(self.level_k)),
                                           L.lib().vidc_nms_segmented_scratch_bytes(B, R)), dtype=torch.uint8, device=device)
        self.props = torch.zeros(B, R, 4, **f32)
        self.prop_scores = torch.zeros(B, R, **f32)
        self.n_props = torch.zeros(B, **i32)
        self.cand_boxes = torch.zeros(B, R, 4, **f32)
        self.cand_scores = torch.zeros(B, R, **f32)
        self.cand_src = torch.zeros(B, R, **i32)
        self.n_cand = torch.zeros(B, **i32)
        self.det_keep = torch.zeros(B, R, **i32)
        self.n_det_keep = torch.zeros(B, **i32)
        self.det_boxes = torch.zeros(B, R, 4, **f32)
        self.det_scores = torch.zeros(B, R, **f32)
        self.n_det = torch.zeros(B, **i32)
        self.pasted = torch.zeros(B, R, H, W, dtype=torch.uint8, device=device)
        self.inst = torch.zeros(B, H, W, dtype=torch.uint8, device=device)
        self.inst_scratch = torch.empty(L.lib().vidc_instance_map_scratch_bytes(B, R, H, W), dtype=torch.uint8, device=device)


class PlaneMaskDetector:
    def __init__(self, confidence_threshold=0.9, device="cuda", nms_inclusive=True):
        self.device = torch.device(device)
        self.model = N.GeneralizedRCNN().to(self.device).eval()
        self.confidence_threshold = float(confidence_threshold)
        self.nms_inclusive = bool(nms_inclusive)
        self._bufs = {}
        self._graphs = {}          # (B, H, W, confidence, inclusive, parameter version) -> hipGraph of the whole detector
        self._eager = False        # inside the capture the programs issue their kernels instead of replaying their own graphs

    def load_state_dict(self, state):
        self._graphs.clear()
        self.model.load_state_dict(state)

    def state_dict(self):
        return self.model.state_dict()

    # ---- stages (each one enqueues on the current stream and returns views of reused buffers) -------------------------------
    def _ctx(self, B, H, W):
        key = (B, H, W)
        if key not in self._bufs:
            Hp, Wp = self.model.padded(H, W)
            self._bufs[key] = _Buffers(B, H, W, Hp, Wp, self.device)
        return self._bufs[key], self.model.programs(B, H, W, self.device)

    def _run(self, prog):
        if self._eager:
            prog.run()
        else:
            N._HipModule._execute(prog)

    def dense(self, images):
        """images (B,3,H,W) RGB in [0,1] on the GPU -> runs the dense program; returns its Program (outputs P2..P5, rpn0..rpn4)."""
        if not images.is_cuda:
            raise RuntimeError("PlaneMaskDetector runs on the GPU only (no CPU fallback)")
        B, _, H, W = images.shape
        _bufs, (dense, _box, _mask) = self._ctx(B, H, W)
        if images.data_ptr() != dense.tensor(dense.inputs["image"]).data_ptr():
            dense.tensor(dense.inputs["image"]).copy_(images, non_blocking=True)
        self._run(dense)
        return dense

    def proposals(self, rpn_maps, B, H, W):
        """rpn_maps: per level a (B,h,w,32) NHWC tensor (3 objectness logits, 12 deltas) -> (props (B,R,4), scores (B,R), n (B,))."""
        bf, _ = self._ctx(B, H, W)
        lib, st = L.lib(), L.current_stream()
        nl = len(bf.level_hw)
        for l, m in enumerate(rpn_maps):
            assert tuple(m.shape) == (B,) + tuple(bf.level_hw[l]) + (32,) and m.is_contiguous()
        maps = (C.c_void_p * nl)(*[m.data_ptr() for m in rpn_maps])
        L.check(lib.vidc_rpn_topk_decode_levels(maps, bf.level_hw_arr.ctypes.data, bf.strides_arr.ctypes.data, bf.anchors_arr.ctypes.data,
                                                bf.level_off.ctypes.data, nl, B, 32, 3, PRE_NMS_TOP_N, H, W, L.ptr(bf.boxes), L.ptr(bf.scores),
                                                bf.slots, st), "rpn_topk_decode_levels")
        L.check(lib.vidc_nms_segmented(L.ptr(bf.boxes), L.ptr(bf.rpn_seg_off), L.ptr(bf.rpn_seg_n), B * nl, max(bf.level_k), RPN_NMS_THRESH,
                                       int(self.nms_inclusive), POST_NMS_TOP_N, L.ptr(bf.keep), L.ptr(bf.n_keep), L.ptr(bf.nms_scratch), st), "nms_segmented")
        L.check(lib.vidc_rpn_select(L.ptr(bf.boxes), L.ptr(bf.scores), L.ptr(bf.keep), L.ptr(bf.n_keep), B, nl, bf.level_off.ctypes.data,
                                    POST_NMS_TOP_N, N.ROI_SLOTS, L.ptr(bf.props), L.ptr(bf.prop_scores), L.ptr(bf.n_props), st), "rpn_select")
        return bf.props, bf.prop_scores, bf.n_props

    def _pool(self, feats, boxes, B, res, out):
        ptrs = (C.c_void_p * 4)(*[f.data_ptr() for f in feats])
        hw = np.array([[f.shape[1], f.shape[2]] for f in feats], dtype=np.int32)
        for f in feats:
            assert f.is_contiguous() and f.shape[3] == 256
        L.check(L.lib().vidc_roi_align_fpn(ptrs, hw.ctypes.data, 4, 256, L.ptr(boxes), B, N.ROI_SLOTS, res, 2, L.ptr(out), L.current_stream()),
                "roi_align_fpn")

    def box_head(self, feats, props, B, H, W):
        """feats: P2..P5 NHWC; props (B,R,4) -> head output (B*R, 32): 2 class logits, 8 box deltas."""
        _bf, (_d, box, _m) = self._ctx(B, H, W)
        self._pool(feats, props, B, 7, box.tensor(box.inputs["pooled"]))
        self._run(box)
        return box.tensor(box.outputs["head"]).view(B * N.ROI_SLOTS, 32)

    def detections(self, head, props, n_props, B, H, W):
        bf, _ = self._ctx(B, H, W)
        lib, st, R = L.lib(), L.current_stream(), N.ROI_SLOTS
        L.check(lib.vidc_det_candidates(L.ptr(head), 32, L.ptr(props), L.ptr(n_props), B, R, H, W, SCORE_THRESH, L.ptr(bf.cand_boxes),
                                        L.ptr(bf.cand_scores), L.ptr(bf.cand_src), L.ptr(bf.n_cand), st), "det_candidates")
        L.check(lib.vidc_nms_segmented(L.ptr(bf.cand_boxes), L.ptr(bf.det_seg_off), L.ptr(bf.det_seg_n), B, R, DET_NMS_THRESH,
                                       int(self.nms_inclusive), 0, L.ptr(bf.det_keep), L.ptr(bf.n_det_keep), L.ptr(bf.nms_scratch), st), "nms_segmented")
        L.check(lib.vidc_det_select(L.ptr(bf.cand_boxes), L.ptr(bf.cand_scores), L.ptr(bf.cand_src), L.ptr(bf.n_cand), L.ptr(bf.det_keep),
                                    L.ptr(bf.n_det_keep), B, R, L.ptr(bf.det_boxes), L.ptr(bf.det_scores), L.ptr(bf.n_det), st), "det_select")
        return bf.det_boxes, bf.det_scores, bf.n_det

    def mask_logits(self, feats, det_boxes, B, H, W):
        _bf, (_d, _b, mask) = self._ctx(B, H, W)
        self._pool(feats, det_boxes, B, 14, mask.tensor(mask.inputs["pooled"]))
        self._run(mask)
        return mask.tensor(mask.outputs["logits"])                     # (B*R, 14, 56, 32)

    def paste(self, logits, det_boxes, n_det, B, H, W):
        bf, _ = self._ctx(B, H, W)
        L.check(L.lib().vidc_mask_paste(L.ptr(logits), 32, 1, MASK_SIZE, L.ptr(det_boxes), L.ptr(n_det), B, N.ROI_SLOTS, H, W, MASK_THRESH,
                                        L.ptr(bf.pasted), L.current_stream()), "mask_paste")
        return bf.pasted

    def instance_map(self, pasted, det_scores, n_det, B, H, W):
        bf, _ = self._ctx(B, H, W)
        L.check(L.lib().vidc_instance_map(L.ptr(pasted), L.ptr(det_scores), L.ptr(n_det), B, N.ROI_SLOTS, H, W, self.confidence_threshold,
                                          MIN_PLANE_FRACTION, L.ptr(bf.inst), L.ptr(bf.inst_scratch), L.current_stream()), "instance_map")
        return bf.inst

    # ---- the reference's entry points ---------------------------------------------------------------------------------------
    def run_on_batch(self, images):
        """(B,3,H,W) in [0,1] on the GPU -> (B,H,W) uint8 instance ids on the GPU (a view of a reused buffer).  Everything after the
        copy of the input -- three programs and sixteen detector launches, all with static shapes and no host synchronisation -- is
        replayed as ONE hipGraph (VIDC_EXEC=graph, the default), captured on first use."""
        if not images.is_cuda:
            raise RuntimeError("PlaneMaskDetector runs on the GPU only (no CPU fallback)")
        if os.environ.get("VIDC_EXEC", "graph") != "graph":
            return self._run_stages(images)
        B, _, H, W = images.shape
        bf, (dense, _b, _m) = self._ctx(B, H, W)
        key = (B, H, W, self.confidence_threshold, self.nms_inclusive, self.model._version)
        graph = self._graphs.get(key)
        inp = dense.tensor(dense.inputs["image"])
        if graph is None:
            self._eager = True
            try:
                self._run_stages(images)                       # warm-up outside capture: kernel attributes, every buffer allocated
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (RCCL's watchdog thread polls events while this thread captures)
                    self._run_stages(inp)
            finally:
                self._eager = False
            self._graphs[key] = graph
        inp.copy_(images, non_blocking=True)
        graph.replay()
        return bf.inst

    def _run_stages(self, images):
        B, _, H, W = images.shape
        dense = self.dense(images)
        feats = [dense.tensor(dense.outputs["P%d" % l]) for l in (2, 3, 4, 5)]
        rpn = [dense.tensor(dense.outputs["rpn%d" % l]) for l in range(5)]
        props, _sc, n_props = self.proposals(rpn, B, H, W)
        head = self.box_head(feats, props, B, H, W)
        det_boxes, det_scores, n_det = self.detections(head, props, n_props, B, H, W)
        logits = self.mask_logits(feats, det_boxes, B, H, W)
        pasted = self.paste(logits, det_boxes, n_det, B, H, W)
        return self.instance_map(pasted, det_scores, n_det, B, H, W)

    def run_on_tensor(self, image_input):
        """COCODemo.run_on_tensor (demo/predictor.py:143-150): (3,H,W) in [0,1] -> (H,W) uint8 numpy instance-id map."""
        return self.run_on_batch(image_input.to(self.device)[None].float())[0].cpu().numpy()
