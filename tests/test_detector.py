"""Plane-mask head kernels: NMS pinned to the reference's own known-answer vectors, ROIAlign against the restatement
of csrc/cpu/ROIAlign_cpu.cpp (parity unpinned by the reference: it has no ROIAlign test and its C++ does not build here)."""
import os

import numpy as np
import pytest
import torch

from oracle import detector_oracle as D

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms_reference_vectors.npz")
gpu = pytest.mark.gpu


def test_nms_oracle_reproduces_reference_vectors():
    """plane_mask_detection/tests/test_nms.py: TestNMS.test_nms_cpu (5 thresholds) and test_nms1_cpu (53 boxes)."""
    f = np.load(GOLDEN)
    for t, g in zip(f["c0_thresholds"], f["c0_keep"]):
        assert np.array_equal(D.nms(f["c0_boxes"], f["c0_scores"], t), g[g >= 0])
    assert np.array_equal(D.nms(f["c1_boxes"], f["c1_scores"], float(f["c1_threshold"])), f["c1_keep"])


def test_nms_oracle_edge_cases():
    assert D.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 9], [20, 20, 29, 29]], np.float32)
    s = np.array([0.5, 0.9, 0.1], np.float32)
    assert list(D.nms(b, s, 1.0)) == [1, 2]                  # identical boxes: IoU == 1 >= 1 suppresses ...
    assert list(D.nms(b, s, 1.0, strict=True)) == [0, 1, 2]  # ... but not with the CUDA kernel's strict >


def test_roi_align_oracle_basics():
    x = np.arange(2 * 3 * 6 * 8, dtype=np.float32).reshape(2, 3, 6, 8)
    # a ROI covering exactly the pixel centres (1,1)..(2,2) with one sample per bin: bilinear at (1.5+.., ...)
    out = D.roi_align_forward(x, np.array([[1, 1.0, 1.0, 3.0, 3.0]], np.float32), 1.0, 2, 2, 1)
    assert out.shape == (1, 3, 2, 2)
    # sample points: y = 1 + (ph + .5), x = 1 + (pw + .5)  -> value = plane(y, x) exactly for a linear image
    for ph in range(2):
        for pw in range(2):
            yy, xx = 1.5 + ph, 1.5 + pw
            assert abs(out[0, 0, ph, pw] - (x[1, 0, 0, 0] + yy * 8 + xx)) < 1e-4
    # malformed (zero-size) ROI is forced to 1x1; a ROI far outside gives zeros
    assert np.isfinite(D.roi_align_forward(x, np.array([[0, 2.0, 2.0, 2.0, 2.0]], np.float32), 1.0, 2, 2, 0)).all()
    assert np.all(D.roi_align_forward(x, np.array([[0, 100.0, 100.0, 120.0, 120.0]], np.float32), 1.0, 2, 2, 2) == 0)


# Hand-computed known answers for the bilinear edge cases of the reference's kernel (csrc/cpu/ROIAlign_cpu.cpp:20-110; the reference has
# no ROIAlign test of its own and its C++ does not build against this PyTorch): feature f[y][x] = 10*y + x on a 4x4 map, ONE sample per
# ROI (pooled 1x1, sampling_ratio 1) at the ROI's centre.  (sample x, sample y) -> expected value:
ROI_KAT = [((1.25, 2.5), 26.25),      # interior: plain bilinear (a linear image is reproduced exactly)
           ((-0.5, -0.5), 0.0),       # -1 <= coordinate <= 0: clamped to 0 (:62-67)  -> f[0][0]
           ((1.5, 3.6), 31.5),        # y_low >= height-1: y_high = y_low = last row, y = 3 (:74-79)  -> 30 + 1.5
           ((3.0, 3.0), 33.0),        # exactly on the last pixel -> f[3][3]
           ((3.7, 3.9), 33.0),        # both coordinates beyond the last pixel centre but inside the map -> f[3][3]
           ((1.5, 4.5), 0.0),         # y > height: the sample contributes nothing (:45-59)
           ((-1.5, 2.0), 0.0)]        # x < -1: nothing


def _kat_inputs():
    f = (10.0 * np.arange(4, dtype=np.float32)[:, None] + np.arange(4, dtype=np.float32)[None, :]).reshape(1, 1, 4, 4)
    rois = np.array([[0, cx - 0.5, cy - 0.5, cx + 0.5, cy + 0.5] for (cx, cy), _ in ROI_KAT], np.float32)      # width = height = 1 -> centre sample
    return f, rois, np.array([v for _, v in ROI_KAT], np.float32)


def test_roi_align_oracle_known_answers():
    f, rois, want = _kat_inputs()
    got = D.roi_align_forward(f, rois, 1.0, 1, 1, 1).reshape(-1)
    assert np.abs(got - want).max() < 1e-5, (got, want)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_roi_align_known_answers():
    from vi_depth_completion_amd import detector
    f, rois, want = _kat_inputs()
    x = np.repeat(f, 32, axis=1)                    # the kernel puts lanes along channels: 32 identical channels
    got = detector.roi_align(torch.from_numpy(x).cuda(), torch.from_numpy(rois).cuda(), (1, 1), 1.0, 1).cpu().numpy()
    assert got.shape == (len(want), 32, 1, 1)
    assert np.abs(got[:, :, 0, 0] - want[:, None]).max() < 1e-5

@gpu
def test_hip_nms_reference_vectors():
    from vi_depth_completion_amd import detector
    f = np.load(GOLDEN)
    for t, g in zip(f["c0_thresholds"], f["c0_keep"]):
        k = detector.nms(torch.from_numpy(f["c0_boxes"]).cuda(), torch.from_numpy(f["c0_scores"]).cuda(), float(t), inclusive=True)
        assert np.array_equal(k.cpu().numpy(), g[g >= 0])
    k = detector.nms(torch.from_numpy(f["c1_boxes"]).cuda(), torch.from_numpy(f["c1_scores"]).cuda(), float(f["c1_threshold"]), inclusive=True)
    assert np.array_equal(k.cpu().numpy(), f["c1_keep"])


@gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 500, 2000, 4500])
@pytest.mark.parametrize("inclusive", [False, True])
def test_hip_nms_random_vs_oracle(n, inclusive):
    """Clustered random boxes (many overlaps, chains across 64-box tiles), both comparison rules; n > 4096 exercises two words per lane."""
    from vi_depth_completion_amd import detector
    rng = np.random.RandomState(n)
    centers = rng.uniform(0, 300, (max(1, n // 12), 2))
    c = centers[rng.randint(0, len(centers), n)] + rng.normal(0, 6, (n, 2))
    wh = rng.uniform(8, 60, (n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / n             # distinct scores: the order is unambiguous
    if n > 2500 and not inclusive:
        pytest.skip("the O(n^2) python oracle takes too long twice")
    want = D.nms(boxes, scores, 0.5, strict=not inclusive)
    got = detector.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.5, inclusive=inclusive).cpu().numpy()
    assert np.array_equal(got, want)
    assert detector.nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), 0.5).numel() == 0


@gpu
@pytest.mark.parametrize("cfg", [(256, 7, 7, 0.25, 2), (64, 14, 14, 0.125, 2), (96, 3, 5, 1.0, 0), (33, 2, 2, 0.5, 3)])
def test_hip_roi_align_vs_oracle(cfg):
    """fp32 tolerance 2e-5 relative to the feature scale: the device may contract a*b+c into FMAs, the restatement does not."""
    from vi_depth_completion_amd import detector
    C, PH, PW, scale, sr = cfg
    rng = np.random.RandomState(C)
    H, W = 24, 31
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    K = 9
    x1, y1 = rng.uniform(-8, W / scale * 0.8, K), rng.uniform(-8, H / scale * 0.8, K)
    rois = np.stack([rng.randint(0, 2, K), x1, y1, x1 + rng.uniform(0, W / scale * 0.6, K), y1 + rng.uniform(0, H / scale * 0.6, K)], axis=1).astype(np.float32)
    rois[0, 1:] = [5, 5, 5, 5]                          # zero-size ROI
    rois[1, 1:] = [W / scale + 50, 0, W / scale + 90, 30]   # entirely outside
    want = D.roi_align_forward(x, rois, scale, PH, PW, sr)
    got = detector.roi_align(torch.from_numpy(x).cuda(), torch.from_numpy(rois).cuda(), (PH, PW), scale, sr).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_segmented_nms_matches_oracle():
    """vidc_nms_segmented: several score-ordered lists of different lengths in one launch set, each against the oracle's greedy NMS."""
    import torch
    from oracle import detector_oracle as DO
    from vi_depth_completion_amd import _lib as L
    rng = np.random.RandomState(11)
    counts = [1000, 333, 64, 65, 1, 50]
    offs = np.concatenate(([0], np.cumsum(counts)))[:-1] + np.arange(len(counts)) * 4      # gaps between the lists
    total = int(offs[-1] + counts[-1])
    boxes = np.zeros((total, 4), np.float32)
    expect = []
    for o, n in zip(offs, counts):
        xy = rng.uniform(0, 300, (n, 2)).astype(np.float32)
        wh = rng.uniform(4, 150, (n, 2)).astype(np.float32)
        boxes[o:o + n] = np.concatenate([xy, xy + wh], 1)
        expect.append(DO.nms(boxes[o:o + n], -np.arange(n, dtype=np.float32), 0.6))          # already in score order
    bd = torch.from_numpy(boxes).cuda()
    so, sn = torch.tensor(offs, dtype=torch.int32).cuda(), torch.tensor(counts, dtype=torch.int32).cuda()
    keep = torch.full((total,), -1, dtype=torch.int32).cuda()
    nk = torch.zeros(len(counts), dtype=torch.int32).cuda()
    scratch = torch.empty(L.lib().vidc_nms_segmented_scratch_bytes(len(counts), 1000), dtype=torch.uint8).cuda()
    L.check(L.lib().vidc_nms_segmented(L.ptr(bd), L.ptr(so), L.ptr(sn), len(counts), 1000, 0.6, 1, 0, L.ptr(keep), L.ptr(nk), L.ptr(scratch),
                                       L.current_stream()), "nms_segmented")
    keep_h, nk_h = keep.cpu().numpy(), nk.cpu().numpy()
    for s, (o, n) in enumerate(zip(offs, counts)):
        assert nk_h[s] == len(expect[s]) and np.array_equal(keep_h[o:o + nk_h[s]], expect[s])
    # max_keep: the first 50 survivors of every list (boxlist_nms(max_proposals=50))
    L.check(L.lib().vidc_nms_segmented(L.ptr(bd), L.ptr(so), L.ptr(sn), len(counts), 1000, 0.6, 1, 50, L.ptr(keep), L.ptr(nk), L.ptr(scratch),
                                       L.current_stream()), "nms_segmented")
    keep_h, nk_h = keep.cpu().numpy(), nk.cpu().numpy()
    for s, (o, n) in enumerate(zip(offs, counts)):
        m = min(50, len(expect[s]))
        assert nk_h[s] == m and np.array_equal(keep_h[o:o + m], expect[s][:m])
