"""Drop-in for the reference's `Warping2DOFAlignment` (networks/warping_2dof_alignment.py:5-24, 108-156, 216-255):
same constructor and method signatures, but each method is two HIP launches (per-sample geometry + fused
gather) through libvidc.so instead of ~520 ATen calls.  No weights; not an nn.Module (like the reference).
"""
import math

import numpy as np
import torch

from .. import _lib as L


class Warping2DOFAlignment:
    def __init__(self, fx=577.87061 * 0.5, fy=577.87061 * 0.5, cx=319.87654 * 0.5, cy=239.87603 * 0.5,
                 align_corners=False, device="cuda"):
        self.device = torch.device(device)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.W, self.H = int(math.ceil(2 * cx)), int(math.ceil(2 * cy))
        k = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
        self.K_host = k.astype(np.float32)
        self.K_inv_host = np.linalg.inv(k).astype(np.float32)       # fp64 inverse cast to fp32, as the reference does
        # grid_sample convention: False = what torch >= 1.3 does when the reference omits the argument (the oracle's
        # behaviour); True = the torch-1.2 behaviour the checkpoints were trained with.
        self.align_corners = bool(align_corners)
        self._kinv_dev = {}

    def kinv(self, device):
        key = str(device)
        if key not in self._kinv_dev:
            self._kinv_dev[key] = torch.from_numpy(self.K_inv_host.reshape(-1).copy()).to(device)
        return self._kinv_dev[key]

    def _params(self, I_g, I_a):
        B = I_g.shape[0]
        g = I_g.reshape(B, 3).contiguous().float()
        a = I_a.reshape(B, 3).contiguous().float()
        p = torch.empty((B, L.WARP_PARAMS), dtype=torch.float32, device=g.device)
        L.check(L.lib().vidc_warp2dof_params(L.ptr(g), L.ptr(a), B, self.fx, self.fy, self.cx, self.cy,
                                             L.ptr(self.kinv(g.device)), self.W, self.H, L.ptr(p), L.current_stream()),
                "warp2dof_params")
        return p

    @staticmethod
    def _require_device(x):
        if not x.is_cuda:
            raise RuntimeError("Warping2DOFAlignment runs on the GPU only (HIP path, no CPU fallback)")

    def warp_with_gravity_center_aligned(self, x, I_g, I_a, interp_mode="bilinear"):
        self._require_device(x)
        if interp_mode != "bilinear":
            raise NotImplementedError("only bilinear interpolation is implemented")
        squeeze = x.dim() == 3
        if squeeze:
            x = x.view(x.shape[0], 1, x.shape[1], x.shape[2])
        assert x.shape[0] == I_g.shape[0]
        assert tuple(x.shape[-2:]) == (self.H, self.W), "image must be %dx%d" % (self.H, self.W)
        x = x.contiguous().float()
        p = self._params(I_g, I_a)
        y = torch.empty_like(x)
        B, Cc = x.shape[0], x.shape[1]
        L.check(L.lib().vidc_warp2dof_fwd(L.ptr(x), L.ptr(p), L.ptr(y), B, Cc, self.H, self.W, self.cx, self.cy,
                                          int(self.align_corners), L.current_stream()), "warp2dof_fwd")
        Cg_H_C = p[:, 0:9].reshape(B, 3, 3).clone()
        return (Cg_H_C, y.view(B, self.H, self.W)) if squeeze else (Cg_H_C, y)

    def inverse_warp_normal_image_with_gravity_center_aligned(self, x, I_g, I_a, normalize=False):
        self._require_device(x)
        assert x.shape[0] == I_g.shape[0] and x.shape[1] == 3
        assert tuple(x.shape[-2:]) == (self.H, self.W)
        x = x.contiguous().float()
        p = self._params(I_g, I_a)
        z = torch.empty_like(x)
        B = x.shape[0]
        L.check(L.lib().vidc_warp2dof_inv_rot_norm(L.ptr(x), L.ptr(p), L.ptr(z), B, self.H, self.W, self.cx, self.cy,
                                                   int(self.align_corners), int(normalize), L.current_stream()),
                "warp2dof_inv_rot_norm")
        return p[:, 0:9].reshape(B, 3, 3).clone(), z
