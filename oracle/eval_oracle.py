"""CPU restatement of the reference's depth evaluation figures: TEST INFRASTRUCTURE (imported only by tests/).
Follows `_network_evaluate` (network_run.py:212-223: mask gt > 0, ratio = max(gt/pred, pred/gt), abs error) and the DEPTH ERROR STATS
line of `evaluate` (network_run.py:396-403), and `SaveDepthsToImage` (network_run.py:42-50).  Parity unpinned by reference tests (the
reference has none); the formulas are restated line by line."""
import numpy as np
import torch


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
