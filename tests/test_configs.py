"""BASELINE.json's own configurations, end to end on the GPU against the CPU oracle (VERDICT r1 `configs_untested`):

configs[2]  "ScanNet-style 640x480 stream, batch=8, full pipeline incl. plane_mask_detection head"
configs[3]  "Azure-Kinect-style 1280x720, batch=32 sharded 4 per GPU": one GPU's share (batch 4) for two consecutive batches of a rank
SURVEY §4(v) / §8e: a frame's output does not depend on which shard (rank r of N) computed it.

Oracle side: oracle/preprocess_oracle.py (DemoDataset's per-frame work, bit-identical to Pillow), oracle/plane_mask_oracle.py
(COCODemo.run_on_tensor), oracle/vidc_oracle.call_cnn (main.py:261-298).  Bars: uint8/float pre-processing outputs and plane-instance
ids are index/byte work -> exact; depth -> RMSE < 1e-3 (north_star)."""
import numpy as np
import pytest
import torch

from oracle import plane_mask_oracle as PM
from oracle import preprocess_oracle as PO
from oracle import vidc_oracle as O
from vi_depth_completion_amd import synthetic as S

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
INTR = O.Intrinsics(202.0, 202.0, 0.5 * 319.87654, 0.5 * 239.87603)


def _oracle_batch(cam):
    """DemoDataset.__getitem__ + default collate for every frame of a raw camera batch, on the CPU."""
    frames = [PO.demo_frame(cam["image_u8"][i].numpy(), cam["gravity_raw"][i], cam["klt_tracks"][i]) for i in range(cam["image_u8"].shape[0])]
    return {k: torch.stack([f[k] for f in frames]) for k in frames[0]}


def _to_dev(batch):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}


def _make_pipeline(seeded_weights, rng):
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    p = DepthCompletionPipeline(enriched_samples=200, rng=rng)
    p.load_state_dicts(seeded_weights["sn"], seeded_weights["dc"])
    p.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    return p


@pytest.fixture(scope="module")
def pipe(seeded_weights):
    return _make_pipeline(seeded_weights, np.random.RandomState(0))


@pytest.fixture(scope="module")
def detector(detector_weights):
    from vi_depth_completion_amd.plane_mask import PlaneMaskDetector
    det = PlaneMaskDetector(device=DEV)
    det.load_state_dict({k: v.to(DEV) for k, v in detector_weights.items()})
    return det


def _has_match(box, score, boxes, scores):
    x0, y0 = np.maximum(box[0], boxes[:, 0]), np.maximum(box[1], boxes[:, 1])
    x1, y1 = np.minimum(box[2], boxes[:, 2]), np.minimum(box[3], boxes[:, 3])
    inter = np.clip(x1 - x0 + 1, 0, None) * np.clip(y1 - y0 + 1, 0, None)
    area = (box[2] - box[0] + 1) * (box[3] - box[1] + 1) + (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1) - inter
    iou = inter / area
    return bool(np.any((iou > 0.98) & (np.abs(scores - score) < 1e-3)))


def _check_preprocessing(dev_batch, ref_batch):
    assert torch.equal(dev_batch["image"].cpu(), ref_batch["image"]), "resize + ToTensor must be Pillow's, bit for bit"
    assert torch.equal(dev_batch["sparse_depth"].cpu(), ref_batch["sparse_depth"])
    assert torch.equal(dev_batch["gravity"].cpu(), ref_batch["gravity"]) and torch.equal(dev_batch["aligned_direction"].cpu(), ref_batch["aligned_direction"])
    assert torch.equal(dev_batch["homogeneous_coordinates"].cpu(), ref_batch["homogeneous_coordinates"])


def test_config2_640x480_batch8_with_plane_head(pipe, detector, detector_weights, seeded_weights, monkeypatch):
    """configs[2] in full: uint8 640x480 frames + VI-SLAM tracks -> device-side pre-processing -> batch 8 -> Mask R-CNN plane head ->
    surface normals -> plane block / enrichment -> depth completion, through `_call_cnn` and through the software-pipelined
    `run_interleaved` (the mode bench.py times).

    The oracle's detector is run on the same frames: its id maps are compared exactly where no detection decision flipped (reported
    per image; random-noise frames put more scores next to the thresholds than camera frames do), and the depth comparison feeds the
    oracle the ids the device produced, so that every image's RANSAC / enrichment draws line up whatever the detector decided."""
    from vi_depth_completion_amd.preprocess import FramePreprocessor
    B = 8
    cam = S.synthetic_camera_batch(B, 480, 640, 1234, frame0=40)
    pre = FramePreprocessor(DEV, in_hw=(480, 640), out_hw=(240, 320))
    dev_batch = pre(cam["image_u8"].to(DEV), cam["gravity_raw"], cam["klt_tracks"])
    ref_batch = _oracle_batch(cam)
    _check_preprocessing(dev_batch, ref_batch)

    saved = pipe.plane_masks_extraction
    try:
        pipe.plane_masks_extraction = detector
        pipe.rng = np.random.RandomState(21)
        got = pipe._call_cnn(dev_batch).cpu()
        ids_dev = [m.copy() for m in pipe._ids_host.numpy()]
        bf, _ = detector._ctx(B, 240, 320)
        n_det = bf.n_det.cpu().numpy()
        dev_scores, dev_boxes = bf.det_scores.cpu().numpy(), bf.det_boxes.cpu().numpy()
        ids_or, agree, flipped, unmatched, or_dets = [], [], [], [], []
        for i in range(B):
            taps = {}
            ids_or.append(PM.run_on_tensor(detector_weights, ref_batch["image"][i], taps=taps))
            so, bo = taps["det_scores"].numpy(), taps["det_boxes"].numpy()
            or_dets.append((so, bo))
            sd, bd = dev_scores[i][:n_det[i]], dev_boxes[i][:n_det[i]]
            # a detection of one side is "matched" when the other side holds the same box (IoU > 0.98) with the same score (1e-3)
            miss = sum(1 for k in range(len(so)) if not len(sd) or not _has_match(bo[k], so[k], bd, sd)) + \
                sum(1 for k in range(len(sd)) if not len(so) or not _has_match(bd[k], sd[k], bo, so))
            unmatched.append(miss)
            agree.append(miss == 0 and len(so) == len(sd))
            flipped.append(int((ids_dev[i] != ids_or[i]).sum()))
        print("configs[2]: per image -- detections (device / oracle lists agree) %s, unmatched detections %s, differing id pixels %s, planes %s"
              % ([int(a) for a in agree], unmatched, flipped, [int(m.max()) for m in ids_dev]))
        # Integer work, asserted exactly: the detector's programs run every conv in exact fp32, direct form, whatever mode the depth path is in
        # (networks/plane_mask_rcnn.detector_arithmetic), so EVERY image must take the oracle's detection decisions -- same boxes, same
        # scores -- and the decisions that follow (score > 0.9, mask > 0.5 per pixel, biggest component, >= 5 % of the image, ids by size)
        # must give the oracle's id map pixel for pixel.  (Round 4 ran the detector in the mixed mode and had to allow one flipped NMS
        # decision on one of these eight noise frames: box pair at IoU 0.4987 / 0.4951 against the 0.5 bar.)
        assert all(agree), (agree, unmatched)
        assert all(f == 0 for f in flipped), flipped
        assert max(int(m.max()) for m in ids_dev) >= 2, "the seeded detector finds planes on these frames"
        # the opt-in mixed-mode detector (VIDC_DETECTOR_PRECISION=mixed): close, not exact -- at most one image with a flipped decision
        from vi_depth_completion_amd.plane_mask import PlaneMaskDetector
        monkeypatch.setenv("VIDC_DETECTOR_PRECISION", "mixed")
        monkeypatch.setenv("VIDC_PRECISION", "mixed")
        detm = PlaneMaskDetector(device=DEV)
        detm.load_state_dict({k: v.to(DEV) for k, v in detector_weights.items()})
        idsm = detm.run_on_batch(dev_batch["image"]).cpu().numpy()
        monkeypatch.delenv("VIDC_DETECTOR_PRECISION")
        monkeypatch.delenv("VIDC_PRECISION")
        differing = [int((idsm[i] != ids_or[i]).sum()) for i in range(B)]
        print("configs[2]: opt-in mixed-mode detector, differing id pixels per image %s" % differing)
        assert sum(1 for d in differing if d) <= 1 and all(d <= 0.03 * 240 * 320 for d in differing), differing
        del detm
        ref = O.call_cnn(seeded_weights["sn"], seeded_weights["dc"], ref_batch, ids_dev, INTR, 200, rng=np.random.RandomState(21))
        rmse = float((got - ref).pow(2).mean().sqrt())
        per_img = (got - ref).pow(2).mean(dim=(1, 2, 3)).sqrt()
        print("configs[2]: depth RMSE vs oracle %.3e (per image max %.3e)" % (rmse, float(per_img.max())))
        assert rmse < 1e-3 and float(per_img.max()) < 1e-3
        # the software-pipelined stream: the same batch twice in a row (two ticks + drain), draws restarted
        pipe.rng = np.random.RandomState(21)
        outs = [o.cpu() for o in pipe.run_interleaved(iter([dev_batch, dev_batch]))]
        assert float((outs[0] - ref).pow(2).mean().sqrt()) < 1e-3
        assert len(outs) == 2 and torch.isfinite(outs[1]).all()
    finally:
        pipe.plane_masks_extraction = saved


def test_two_lanes_with_the_plane_head_are_bit_identical(pipe, detector):
    """The mode `bench.py --plane-head` times: `run_interleaved(lanes=2)` with the Mask R-CNN plane head in the loop (its id maps travel
    device -> host -> plane block every frame, through a staging buffer the lanes share) and the enrichment wait of a frame deferred to
    the next visit.  Six batch-2 frames: every depth map bit-identical to the one-lane stream's."""
    frames = [_to_dev(S.synthetic_batch(2, 240, 320, 1234, frame0=200 + 2 * i)) for i in range(6)]
    saved = pipe.plane_masks_extraction
    try:
        pipe.plane_masks_extraction = detector
        runs = []
        for lanes in (1, 2):
            pipe.rng = np.random.RandomState(17)
            runs.append([o.cpu() for o in pipe.run_interleaved(iter(frames), lanes=lanes)])
        assert len(runs[0]) == len(runs[1]) == 6
        for f, (a, b) in enumerate(zip(*runs)):
            assert torch.equal(a, b), "frame %d differs between one and two lanes" % f
        assert not torch.equal(runs[0][0], runs[0][1])
    finally:
        pipe.plane_masks_extraction = saved


def test_config3_1280x720_share_of_one_gpu(pipe, seeded_weights):
    """configs[3]: 1280x720 stream, global batch 32 = 4 per GPU x 8.  Rank 5 of 8 takes global batches 5 and 13 (frames round-robin by
    batch, bench.py): uint8 1280x720 -> device-side pre-processing -> batch 4 -> the pipeline (plane mask fixed, like the bench line
    of this configuration), via run_interleaved.  Every frame against the oracle."""
    from vi_depth_completion_amd.preprocess import FramePreprocessor
    B, rank, world = 4, 5, 8
    pre = FramePreprocessor(DEV, in_hw=(720, 1280), out_hw=(240, 320))
    cams = [S.synthetic_camera_batch(B, 720, 1280, 1234, frame0=(rank + j * world) * B) for j in range(2)]
    dev_batches = [pre(c["image_u8"].to(DEV), c["gravity_raw"], c["klt_tracks"]) for c in cams]
    ref_batches = [_oracle_batch(c) for c in cams]
    for d, r in zip(dev_batches, ref_batches):
        _check_preprocessing(d, r)
    pipe.rng = np.random.RandomState(33)
    outs = [o.cpu() for o in pipe.run_interleaved(iter(dev_batches))]
    rng = np.random.RandomState(33)
    masks = [S.plane_id_map(240, 320)] * B
    for j in range(2):
        ref = O.call_cnn(seeded_weights["sn"], seeded_weights["dc"], ref_batches[j], masks, INTR, 200, rng=rng)
        per_img = (outs[j] - ref).pow(2).mean(dim=(1, 2, 3)).sqrt()
        print("configs[3] batch %d of rank %d: depth RMSE per image %s" % (j, rank, ["%.2e" % float(v) for v in per_img]))
        assert float(per_img.max()) < 1e-3


def test_frame_output_does_not_depend_on_the_shard(seeded_weights):
    """SURVEY §8e determinism requirement: frame f computed by rank 0 of 1 (which runs frames 0,1,2,3) and by rank 1 of 2 (frames
    1,3) is bit-identical -- two pipeline objects on one GPU, each frame with its own generator so that the shard's draw history does
    not enter.  In `run_interleaved` a frame shares its launches with a different neighbour in the two shards (its surface-normal
    pass rides with frame f-1's depth pass in one shard and with frame f-2's in the other); the groups of a launch do not interact."""
    frames = {f: _to_dev(S.synthetic_batch(1, 240, 320, 1234, frame0=f)) for f in range(4)}

    def run_shard(rank, world, n):
        p = _make_pipeline(seeded_weights, np.random.RandomState(0))
        ids = list(range(rank, n, world))
        seq = {}
        for f in ids:
            p.rng = np.random.RandomState(1000 + f)
            seq[f] = p._call_cnn(frames[f]).cpu()

        # one item per launch, one lane (frame f draws from its own generator: `frame_rng`, whatever the scheduler pulls ahead)
        inter = dict(zip(ids, (o.cpu() for o in p.run_interleaved(iter([frames[f] for f in ids]), frame_rng=lambda i: np.random.RandomState(1000 + ids[i])))))
        # the mode bench.py times: two lanes, two consecutive frames of the shard per launch (rank 0 of 1 pairs (0,1) (2,3); rank 1 of 2
        # pairs (1,3): frame 1 changes batch slot and partner, frame 3 its partner)
        paired = dict(zip(ids, (o.cpu() for o in p.run_interleaved(iter([frames[f] for f in ids]), lanes=2, frames_per_launch=2,
                                                                   frame_rng=lambda i: np.random.RandomState(1000 + ids[i])))))
        return seq, inter, paired

    seq_a, int_a, pair_a = run_shard(0, 1, 4)
    seq_b, int_b, pair_b = run_shard(1, 2, 4)
    for f in (1, 3):
        assert torch.equal(seq_a[f], seq_b[f]), "frame %d differs between shards (sequential path)" % f
        assert torch.equal(int_a[f], int_b[f]), "frame %d differs between shards (interleaved path)" % f
        assert torch.equal(pair_a[f], pair_b[f]), "frame %d differs between shards (two frames per launch)" % f
        assert float((pair_a[f] - seq_a[f]).pow(2).mean().sqrt()) < 1e-3
    assert not torch.equal(seq_a[1], seq_a[3]) and not torch.equal(pair_a[1], pair_a[3])
