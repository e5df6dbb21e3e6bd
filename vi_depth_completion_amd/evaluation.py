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


NORMAL_NAMES = ("5deg", "7.5deg", "11.25deg", "22.5deg", "30deg")


class NormalErrorStats:
    """NORMAL ERROR STATS of the reference's `evaluate()` (network_run.py:389-397) from `_network_evaluate`'s angle errors
    (:204-214), on the device.  Mean / RMSE / the five threshold percentages come from 8 fp64 running sums like the depth figures.
    The MEDIAN (np.median over the concatenated errors of the whole test set) is exact too: every batch leaves its per-pixel errors
    on the device, and `result()` selects the middle order statistic(s) with two 16-bit radix-histogram passes over them -- the
    histograms are integer counts, so a frame-sharded evaluation all-reduces 256 KB per pass instead of gathering the errors."""

    def __init__(self, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("NormalErrorStats needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device(device)
        self.totals = torch.zeros(N_STATS, dtype=torch.float64, device=self.device)
        self.errors = []          # per-batch fp32 (B*HW,) error arrays (all-ones bits where the mask is off)
        self._scratch = None

    def update(self, pred_normals, normals_gt, mask):
        """pred_normals, normals_gt: (B,3,H,W) fp32 GPU tensors (normalised here, like network_run.py:205,207); mask: (B,H,W), valid > 0."""
        if not (pred_normals.is_cuda and normals_gt.is_cuda and mask.is_cuda):
            raise RuntimeError("NormalErrorStats.update takes GPU tensors")
        p, g, m = pred_normals.contiguous().float(), normals_gt.contiguous().float(), mask.contiguous().float()
        B, _, H, W = p.shape
        assert g.shape == p.shape and m.numel() == B * H * W
        need = L.lib().vidc_depth_metrics_scratch_bytes(B * H * W)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        err = torch.empty(B * H * W, dtype=torch.float32, device=self.device)
        L.check(L.lib().vidc_normal_metrics(L.ptr(p), L.ptr(g), L.ptr(m), B, H * W, L.ptr(err), L.ptr(self.totals), 1, L.ptr(self._scratch),
                                            L.current_stream()), "normal_metrics")
        self.errors.append(err)
        return err

    def _hist(self, hi_filter):
        h = torch.zeros(65536, dtype=torch.int32, device=self.device)
        for e in self.errors:
            L.check(L.lib().vidc_hist_u16(L.ptr(e), e.numel(), hi_filter, L.ptr(h), L.current_stream()), "hist_u16")
        h = h.to(torch.int64)
        all_reduce_totals(h)
        return h.cpu().numpy()

    def order_statistics(self, ranks):
        """Exact k-th smallest errors (0-based ranks over all ranks' valid pixels), as float32."""
        hi = self._hist(-1)
        chi = np.cumsum(hi)
        out = {}
        for hb in sorted({int(np.searchsorted(chi, k, side="right")) for k in ranks}):
            lo = self._hist(hb)
            clo = np.cumsum(lo)
            below = int(chi[hb - 1]) if hb > 0 else 0
            for k in ranks:
                if int(np.searchsorted(chi, k, side="right")) == hb:
                    lb = int(np.searchsorted(clo, k - below, side="right"))
                    out[k] = np.array([(hb << 16) | lb], dtype=np.uint32).view(np.float32)[0]
        return [out[k] for k in ranks]

    def all_reduce(self):
        all_reduce_totals(self.totals)
        self._reduced = True
        return self

    def result(self):
        t = self.totals.cpu().numpy()
        n = int(t[0])
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not getattr(self, "_reduced", False):
            raise RuntimeError("call all_reduce() before result() in a sharded evaluation (the median passes are collective)")
        if n <= 0:
            return None
        a, b = self.order_statistics([(n - 1) // 2, n // 2])
        out = {"n": n, "Mean": float(t[1] / n), "Median": float(np.float32(0.5) * (np.float32(a) + np.float32(b))) if a != b else float(a),
               "Rmse": float(np.sqrt(t[2] / n))}
        for name, c in zip(NORMAL_NAMES, t[3:8]):
            out[name] = 100.0 * float(c) / n
        return out


def all_reduce_totals(totals):
    """In-place SUM of the 8 running totals over ranks: the whole cross-GPU traffic of a frame-sharded evaluation (64 bytes)."""
    import torch.distributed as dist
    from .sharding import collectives_active
    if collectives_active():
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
