"""Device-side frame pre-processing: the per-frame work of the reference's `DemoDataset.__getitem__` (dataset.py:461-520) behind the
same output dictionary, so a camera stream can be fed to `DepthCompletionPipeline` without a PIL/Python DataLoader per frame.

    pre = FramePreprocessor(device)                       # tables for 640x480 -> 320x240, DemoDataset's intrinsics
    batch = pre(image_u8, gravity_raw, klt_tracks)        # image_u8: (B,480,640,3) uint8; gravity_raw: (B,3); klt_tracks: list of (N_i,4)
    depth = pipeline._call_cnn(batch)

What runs where: the resize (+ToTensor) and the sparse-point rasterisation are HIP kernels (csrc/preprocess.hip, bit-identical to
Pillow / to the reference's float64 loop); the gravity sign flip and alignment rule are three floats per frame and stay on the host
in the reference's own torch-CPU arithmetic (dataset.py:472-483); the homogeneous grid is a constant (dataset.py:34-42).
There is no CPU fallback for the kernels."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

DEMO_FC = (202.9953, 202.9540)       # dataset.py:456-457: "compensates for both cropping and scaling"
DEMO_CC = (159.7645, 122.0951)


def resize_tables(in_size, out_size):
    """(bounds int32 [out,2], coeffs int32 [out,ksize]) of Pillow's bilinear resample for one axis (host computation in libvidc)."""
    lib = L.lib()
    ks = C.c_int(0)
    L.check(lib.vidc_resize_coeffs(in_size, out_size, None, None, 0, C.byref(ks)), "resize_coeffs")
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ks.value), np.int32)
    L.check(lib.vidc_resize_coeffs(in_size, out_size, bounds.ctypes.data, coeffs.ctypes.data, coeffs.size, C.byref(ks)), "resize_coeffs")
    return bounds, coeffs


def nearest_table(in_size, out_size):
    """int32 [out]: source index of every output coordinate of Pillow's NEAREST resize along one axis (host computation in libvidc)."""
    t = np.zeros(out_size, np.int32)
    L.check(L.lib().vidc_nearest_table(in_size, out_size, t.ctypes.data), "nearest_table")
    return t


def gravity_and_alignment(gravity_raw):
    """dataset.py:472-483, host, torch CPU fp32 like the reference."""
    g = torch.tensor(np.asarray(gravity_raw, dtype=np.float64), dtype=torch.float)
    g[1] = -g[1]
    g[2] = -g[2]
    psi = g[1] * g[1] + g[2] * g[2]
    if psi < 1e-4:
        a = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float)
    else:
        pitch = torch.atan2(g[2], g[1])
        if torch.cos(pitch) > 0.707:
            a = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float)
        else:
            a = torch.tensor([0.0, torch.cos(pitch), torch.sin(pitch)], dtype=torch.float)
    return g, a


def homogeneous_coordinates(fc, cc, W, H):
    """dataset.py:34-42 (float64 numpy, then float32)."""
    hom = np.zeros((H, W, 3))
    hom[:, :, 2] = 1
    xx, yy = np.meshgrid(np.arange(W), np.arange(H))
    hom[:, :, 0] = (xx - cc[0]) / fc[0]
    hom[:, :, 1] = (yy - cc[1]) / fc[1]
    return torch.from_numpy(hom.astype(np.float32))


class FramePreprocessor:
    def __init__(self, device="cuda", in_hw=(480, 640), out_hw=(240, 320), fc=DEMO_FC, cc=DEMO_CC):
        if not torch.cuda.is_available():
            raise RuntimeError("FramePreprocessor needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device(device)
        self.in_hw, self.out_hw, self.fc, self.cc = tuple(in_hw), tuple(out_hw), tuple(fc), tuple(cc)
        bx, kx = resize_tables(in_hw[1], out_hw[1])
        by, ky = resize_tables(in_hw[0], out_hw[0])
        self.ksx, self.ksy = kx.shape[1], ky.shape[1]
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.bx, self.kx, self.by, self.ky = up(bx), up(kx), up(by), up(ky)
        self.homogeneous = homogeneous_coordinates(fc, cc, out_hw[1], out_hw[0]).to(self.device)
        self.nx, self.ny = up(nearest_table(in_hw[1], out_hw[1])), up(nearest_table(in_hw[0], out_hw[0]))

    def resize(self, image_u8):
        """(B,H,W,C) uint8 on the device -> (B,C,Ho,Wo) float32 = ToTensor(PIL bilinear resize)."""
        if not image_u8.is_cuda or image_u8.dtype != torch.uint8:
            raise RuntimeError("resize() takes a uint8 GPU tensor (B,H,W,C)")
        x = image_u8.contiguous()
        B, H, W, Cc = x.shape
        assert (H, W) == self.in_hw, "tables were built for %s input, got %s" % (self.in_hw, (H, W))
        Ho, Wo = self.out_hw
        y = torch.empty((B, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
        L.check(L.lib().vidc_resize_bilinear_u8_to_chw(L.ptr(x), L.ptr(y), B, H, W, Cc, Ho, Wo, L.ptr(self.bx), L.ptr(self.kx), self.ksx,
                                                       L.ptr(self.by), L.ptr(self.ky), self.ksy, L.current_stream()), "resize")
        return y

    def gt_depth(self, depth_u16, millimetres_per_metre=1000.0):
        """(B,H,W) uint16 millimetre depth maps on the device -> (B,1,Ho,Wo) float32 metres: the ground truth of the training and
        evaluation streams, `Image.open(d).convert('F').resize((320, 240), resample=Image.NEAREST)` then `/ 1000.0`
        (dataset.py:283-286), bit for bit -- `sample['depth']` of network_run.py:165-172, 216-225 without a PIL pass per frame."""
        if not depth_u16.is_cuda or depth_u16.dtype not in (torch.uint16, torch.int16):
            raise RuntimeError("gt_depth() takes a uint16 GPU tensor (B,H,W) (a 16-bit depth PNG as the camera driver delivers it)")
        x = depth_u16.contiguous()
        B, H, W = x.shape
        assert (H, W) == self.in_hw, "tables were built for %s input, got %s" % (self.in_hw, (H, W))
        Ho, Wo = self.out_hw
        y = torch.empty((B, 1, Ho, Wo), dtype=torch.float32, device=x.device)
        L.check(L.lib().vidc_resize_nearest_u16_depth(L.ptr(x), L.ptr(y), B, H, W, Ho, Wo, L.ptr(self.nx), L.ptr(self.ny), float(millimetres_per_metre),
                                                      L.current_stream()), "gt_depth")
        return y

    def _upload(self, t):
        """Small host tensor -> fresh device tensor through (cached) pinned memory: a copy out of pageable memory would block the host
        until the device has caught up with everything queued before it -- a whole tick of the stream being pre-processed for."""
        p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        p.copy_(t)
        return p.to(self.device, non_blocking=True)

    def rasterize(self, klt_tracks):
        """list of B arrays (N_i, 4) = (id, X, Y, Z) -> (B,1,Ho,Wo) float32 sparse depth."""
        B = len(klt_tracks)
        Ho, Wo = self.out_hw
        rows = [np.atleast_2d(np.asarray(t, dtype=np.float64)).reshape(-1, 4) if np.size(t) else np.zeros((0, 4)) for t in klt_tracks]
        offs = np.zeros(B + 1, np.int32)
        offs[1:] = np.cumsum([r.shape[0] for r in rows])
        allr = np.concatenate(rows) if offs[-1] else np.zeros((1, 4))
        tr = self._upload(torch.from_numpy(np.ascontiguousarray(allr)))
        of = self._upload(torch.from_numpy(offs))
        d = torch.empty((B, 1, Ho, Wo), dtype=torch.float32, device=self.device)
        L.check(L.lib().vidc_rasterize_sparse_depth(L.ptr(tr), L.ptr(of), B, self.fc[0], self.fc[1], self.cc[0], self.cc[1], L.ptr(d), Ho, Wo,
                                                    L.current_stream()), "rasterize")
        self._keep = (tr, of)
        return d

    def __call__(self, image_u8, gravity_raw, klt_tracks, depth_u16=None):
        """The collated batch dictionary of DemoDataset (dataset.py:515-520), on the device; with `depth_u16` also the 'depth' entry of
        the Azure / ScanNet loaders (dataset.py:283-286, 349)."""
        img = image_u8 if torch.is_tensor(image_u8) else torch.from_numpy(np.ascontiguousarray(image_u8))
        img = img.to(self.device, non_blocking=True)
        B = img.shape[0]
        ga = [gravity_and_alignment(g) for g in np.asarray(gravity_raw, dtype=np.float64).reshape(B, 3)]
        out = {"image": self.resize(img), "sparse_depth": self.rasterize(klt_tracks),
               "gravity": self._upload(torch.stack([g for g, _ in ga])), "aligned_direction": self._upload(torch.stack([a for _, a in ga])),
               "homogeneous_coordinates": self.homogeneous.unsqueeze(0).expand(B, -1, -1, -1)}
        if depth_u16 is not None:
            d16 = depth_u16 if torch.is_tensor(depth_u16) else torch.from_numpy(np.ascontiguousarray(depth_u16))
            out["depth"] = self.gt_depth(d16.to(self.device, non_blocking=True))
        return out
