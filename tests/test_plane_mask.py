"""Plane-mask detector.  CPU: the oracle restatement (oracle/plane_mask_oracle.py) against golden vectors produced
by the reference itself (oracle/tools/make_golden_plane_mask.py).  GPU: the HIP path against the oracle, stage by stage with the
oracle's inputs ("teacher forcing": discrete decisions -- top-k, NMS, thresholds -- are compared on identical inputs) and end to end."""
import os

import numpy as np
import pytest
import torch

from oracle import plane_mask_oracle as PM
from vi_depth_completion_amd import synthetic as S

torch.set_grad_enabled(False)


@pytest.fixture(scope="session")
def detector_weights(golden_dir):
    man = np.load(os.path.join(golden_dir, "plane_mask_manifest.npz"))
    shapes, a = {}, 0
    for k, s in zip(man["keys"], man["shapes"]):
        shp = eval(s)
        if "anchor_generator" in k:
            shapes[str(k)] = torch.from_numpy(man["anchors"][a:a + shp[0]].copy())
            a += shp[0]
        else:
            shapes[str(k)] = torch.empty(shp, device="meta")
    return S.seeded_detector_state_dict(shapes, 1234)


@pytest.fixture(scope="session")
def oracle_runs(golden_dir, detector_weights):
    """name -> (golden npz, oracle taps, oracle instance map)"""
    out = {}
    for name in ("demo", "synthetic"):
        g = np.load(os.path.join(golden_dir, "plane_mask_%s.npz" % name))
        taps = {}
        inst = PM.run_on_tensor(detector_weights, torch.from_numpy(g["image"]), taps=taps)
        out[name] = (g, taps, inst)
    return out


def test_anchor_buffers_match_reference(golden_dir):
    man = np.load(os.path.join(golden_dir, "plane_mask_manifest.npz"))
    mine = torch.cat([PM.cell_anchors(s, z) for s, z in zip(PM.ANCHOR_STRIDES, PM.ANCHOR_SIZES)]).numpy()
    assert np.array_equal(mine, man["anchors"])


@pytest.mark.parametrize("name", ["demo", "synthetic"])
def test_oracle_matches_reference_golden(oracle_runs, name):
    """Every stage of the restatement against the reference's own outputs.  Floats: 1e-5 (same torch-CPU kernels; observed 0);
    discrete results (detections kept, pasted masks, instance-id map): exact."""
    g, t, inst = oracle_runs[name]
    for l in range(5):
        f = t["feats"][l][0]
        assert np.abs(f[::16, ::3, ::3].numpy() - g["feat%d_probe" % l]).max() < 1e-5
        assert abs(float(f.double().abs().sum()) - g["feat%d_sum" % l][1]) < 1e-6 * g["feat%d_sum" % l][1]
        lg, dl = t["rpn_logits"][l][0], t["rpn_deltas"][l][0]
        if l < 2:
            lg, dl = lg[:, ::4, ::4], dl[:, ::4, ::4]
        assert np.abs(lg.numpy() - g["rpn_logits%d" % l]).max() < 1e-5 and np.abs(dl.numpy() - g["rpn_deltas%d" % l]).max() < 1e-5
    assert np.abs(t["proposals"].numpy() - g["proposals"]).max() < 1e-4 and np.abs(t["objectness"].numpy() - g["objectness"]).max() < 1e-6
    assert np.abs(t["class_logits"].numpy() - g["class_logits"]).max() < 1e-5
    assert np.abs(t["det_boxes"].numpy() - g["det_boxes"]).max() < 1e-4 and np.array_equal(t["det_labels"].numpy(), g["det_labels"])
    assert np.abs(t["det_scores"].numpy() - g["det_scores"]).max() < 1e-6
    assert np.abs(t["mask_prob"].numpy() - g["mask_prob"].astype(np.float32)).max() < 1e-3          # stored as fp16
    assert np.array_equal(np.packbits(t["pasted"].numpy().astype(bool), axis=-1), g["pasted_packed"])
    assert np.array_equal(inst, g["instance_map"]) and inst.max() >= 2


def test_fast_nms_equals_pinned_nms():
    """plane_mask_oracle.nms (vectorised rows) takes the same decisions as detector_oracle.nms, which is pinned to the reference's
    own NMS test vectors."""
    from oracle import detector_oracle as DO
    rng = np.random.RandomState(3)
    for n in (1, 17, 300):
        xy = rng.uniform(0, 200, (n, 2)).astype(np.float32)
        wh = rng.uniform(5, 120, (n, 2)).astype(np.float32)
        boxes = np.concatenate([xy, xy + wh], 1)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        for thr in (0.3, 0.5, 0.7):
            assert np.array_equal(PM.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy(), DO.nms(boxes, scores, thr))
