"""CPU restatement of the reference's evaluation figures: TEST INFRASTRUCTURE (imported only by tests/).
Follows `_network_evaluate` (network_run.py:198-225: normals -- F.normalize both, clamped dot, acos / pi * 180 on mask > 0; depth -- mask
gt > 0, ratio = max(gt/pred, pred/gt), abs error), the NORMAL / DEPTH ERROR STATS lines of `evaluate` (network_run.py:389-403) and
`SaveDepthsToImage` (network_run.py:42-50).
PINNED against the reference itself: oracle/tools/make_golden_eval.py imports network_run.py, runs `_network_evaluate` and `evaluate` on
seeded inputs and stores the error arrays and the figures the reference logged (tests/golden/eval_reference.npz;
tests/test_oracle_golden.py::test_eval_oracle_matches_reference)."""
import numpy as np
import torch
import torch.nn.functional as F


def normal_error_array(pred_normals, normals_gt, mask):
    """network_run.py:204-214.  pred, gt: (B,3,H,W); mask: (B,H,W).  Returns the float32 angle errors (degrees) of the valid pixels."""
    pred = F.normalize(pred_normals)
    m = (mask > 0)[:, None, :, :]
    gt = F.normalize(normals_gt)
    dot = torch.clamp(torch.sum(pred * gt, dim=1), min=-1.0, max=1.0)
    ang = torch.acos(dot) / np.pi * 180
    return ang.numpy()[m[:, 0].numpy() > 0]


def normal_error_stats(err):
    """The NORMAL ERROR STATS figures (network_run.py:389-397) of the concatenated float32 error array."""
    n = err.shape[0]
    return {"n": n, "Mean": float(np.average(err)), "Median": float(np.median(err)), "Rmse": float(np.sqrt(np.sum(err * err) / n)),
            "5deg": 100 * np.sum(err < 5) / n, "7.5deg": 100 * np.sum(err < 7.5) / n, "11.25deg": 100 * np.sum(err < 11.25) / n,
            "22.5deg": 100 * np.sum(err < 22.5) / n, "30deg": 100 * np.sum(err < 30) / n}


def depth_error_arrays(pred, gt):
    mask = (gt > 0).numpy()
    ratio = torch.max(gt / pred, pred / gt).numpy()[mask]
    abs_err = (gt - pred).abs().numpy()[mask]
    return ratio, abs_err


def depth_error_stats(ratio, abs_err):
    n = ratio.shape[0]
    return {"n": n, "MAD": float(np.mean(abs_err.astype(np.float64))), "RMSE": float(np.sqrt(np.mean(abs_err.astype(np.float64) ** 2))),
            "1.05": 100 * np.sum(ratio < 1.05) / n, "1.10": 100 * np.sum(ratio < 1.10) / n, "1.25": 100 * np.sum(ratio < 1.25) / n,
            "1.25^2": 100 * np.sum(ratio < 1.25 ** 2) / n, "1.25^3": 100 * np.sum(ratio < 1.25 ** 3) / n}


def depth_to_mm(depth_np):
    return (depth_np * 1000).astype(np.uint32)
