"""Evaluation statistics on the device: the DEPTH ERROR STATS of the reference's `evaluate()` (network_run.py:349-403) without copying
two full-resolution error maps per batch to the host (`_network_evaluate`, network_run.py:198-225).  Every batch is reduced on the GPU
to 8 fp64 sums (csrc/metrics.hip); `DepthErrorStats.result()` turns the running totals into the reference's figures, and
`all_reduce()` sums them over ranks (RCCL when the process group is "nccl") for frame-sharded evaluation."""
import numpy as np
import torch

from . import _lib as L

N_STATS = 8
RATIO_NAMES = ("1.05", "1.10", "1.25", "1.25^2", "1.25^3")


class DepthErrorStats:
    def __init__(self, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("DepthErrorStats needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device(device)
        self.totals = torch.zeros(N_STATS, dtype=torch.float64, device=self.device)
        self._scratch = None

    def update(self, pred_depths, depths_gt):
        """pred_depths, depths_gt: same-shape fp32 GPU tensors (B,1,H,W); pixels with gt > 0 count (network_run.py:218-223)."""
        if not (pred_depths.is_cuda and depths_gt.is_cuda):
            raise RuntimeError("DepthErrorStats.update takes GPU tensors")
        p, g = pred_depths.contiguous().float(), depths_gt.contiguous().float()
        assert p.shape == g.shape
        n = p.numel()
        need = L.lib().vidc_depth_metrics_scratch_bytes(n)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        L.check(L.lib().vidc_depth_metrics(L.ptr(p), L.ptr(g), n, L.ptr(self.totals), 1, L.ptr(self._scratch), L.current_stream()), "depth_metrics")

    def all_reduce(self):
        """Sum the running totals over the ranks of the default process group (no-op without one)."""
        all_reduce_totals(self.totals)
        return self

    def result(self):
        return stats_to_figures(self.totals.cpu().numpy())


def all_reduce_totals(totals):
    """In-place SUM of the 8 running totals over ranks: the whole cross-GPU traffic of a frame-sharded evaluation (64 bytes)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    return totals


def stats_to_figures(t):
    """The numbers of the reference's 'DEPTH ERROR STATS' log line (network_run.py:396-403) from the 8 sums."""
    n = float(t[0])
    if n <= 0:
        return None
    out = {"n": int(n), "MAD": float(t[1] / n), "RMSE": float(np.sqrt(t[2] / n))}
    for name, c in zip(RATIO_NAMES, t[3:8]):
        out[name] = 100.0 * float(c) / n
    return out


def depth_to_mm_u32(depth):
    """`(depths * 1000).astype(np.uint32)` of SaveDepthsToImage (network_run.py:42-50) on the device; returns an int32-typed GPU tensor
    holding the uint32 bits (torch has no uint32 arithmetic; `.cpu().numpy().view(np.uint32)` gives the image PIL writes)."""
    d = depth.contiguous().float()
    out = torch.empty(d.shape, dtype=torch.int32, device=d.device)
    L.check(L.lib().vidc_depth_to_mm_u32(L.ptr(d), L.ptr(out), d.numel(), L.current_stream()), "depth_to_mm")
    return out
