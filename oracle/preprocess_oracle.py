"""CPU restatement of the reference's per-frame host pre-processing: TEST INFRASTRUCTURE -- only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path (vi_depth_completion_amd/preprocess.py +
csrc/preprocess.hip) never does.

Follows `DemoDataset.__getitem__` (dataset.py:461-520) and `generate_image_homogeneous_coordinates` (dataset.py:34-42):

  * RGB 640x480 -> `Image.resize((320, 240), resample=Image.BILINEAR)` (dataset.py:470) -> `ToTensor()` (uint8 HWC / 255 -> float CHW).
    The arithmetic lives in a third-party dependency that is NOT vendored in the reference: Pillow (the reference pins none; this
    container has Pillow 12.2.0).  `pil_bilinear_resize_u8` restates Pillow's published algorithm (src/libImaging/Resample.c:
    precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; triangle filter of support 1.0 scaled by
    the down-scale factor, 22-bit fixed-point coefficients, horizontal pass then vertical pass with an 8-bit intermediate image) and
    is pinned bit-exactly against Pillow itself (tests/test_preprocess.py) and against the reference's own DemoDataset output
    (tests/golden/preprocess_demo_000000.npz, written by oracle/tools/make_golden_preprocess.py).
  * gravity: sign flip of y, z and the alignment rule (dataset.py:472-483).
  * sparse depth: KLT tracks (id, X, Y, Z) -> u = X/Z, v = Y/Z, col = int(fx*u + cx), row = int(fy*v + cy) in float64, depth = Z,
    later tracks overwrite earlier ones (dataset.py:495-510).
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2          # Resample.c
DEMO_FC = (202.9953, 202.9540)       # dataset.py:456-457
DEMO_CC = (159.7645, 122.0951)


def resample_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1.0) filter over the whole input range.
    Returns (bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out, ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int): truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    scaled = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS))
    return bounds, np.trunc(scaled).astype(np.int32)


def _resample_axis0(img, out_size):
    b, k = resample_coeffs(img.shape[0], out_size)
    src = img.astype(np.int64)
    out = np.zeros((out_size,) + img.shape[1:], np.int64)
    for xx in range(out_size):
        xmin, n = b[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(k[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)        # clip8
    return out.astype(np.uint8)


def pil_bilinear_resize_u8(img_hwc, out_w, out_h):
    """uint8 (H, W, C) -> uint8 (out_h, out_w, C), bit-identical to PIL.Image.resize((out_w, out_h), Image.BILINEAR)."""
    t = np.moveaxis(_resample_axis0(np.moveaxis(img_hwc, 1, 0), out_w), 0, 1)      # horizontal pass first
    return _resample_axis0(t, out_h)


def gravity_and_alignment(gravity_raw):
    """dataset.py:472-483 with torch CPU fp32 ops, like the reference."""
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


def rasterize_sparse_depth(klt_tracks, H=240, W=320, fc=DEMO_FC, cc=DEMO_CC):
    """dataset.py:495-510: rows (id, X, Y, Z) in float64; returns float32 (1, H, W)."""
    out = torch.zeros(1, H, W)
    tr = np.atleast_2d(np.asarray(klt_tracks, dtype=np.float64))
    if tr.size == 0:
        return out
    for i in range(tr.shape[0]):
        u = tr[i, 1] / tr[i, 3]
        v = tr[i, 2] / tr[i, 3]
        col = int(fc[0] * u + cc[0])
        row = int(fc[1] * v + cc[1])
        if 0 <= row < H and 0 <= col < W:
            out[0, row, col] = tr[i, 3]
    return out


def homogeneous_coordinates(fc, cc, W, H):
    """dataset.py:34-42."""
    hom = np.zeros((H, W, 3))
    hom[:, :, 2] = 1
    xx, yy = np.meshgrid(np.arange(W), np.arange(H))
    hom[:, :, 0] = (xx - cc[0]) / fc[0]
    hom[:, :, 1] = (yy - cc[1]) / fc[1]
    return torch.from_numpy(hom.astype(np.float32))


def demo_frame(image_u8_hwc, gravity_raw, klt_tracks, W=320, H=240):
    """The dictionary DemoDataset.__getitem__ returns (without the file name), from the raw file contents."""
    img = pil_bilinear_resize_u8(np.asarray(image_u8_hwc), W, H)
    color = torch.from_numpy(img).permute(2, 0, 1).float().div(255)        # transforms.ToTensor
    g, a = gravity_and_alignment(gravity_raw)
    return {"image": color, "sparse_depth": rasterize_sparse_depth(klt_tracks, H, W), "gravity": g, "aligned_direction": a,
            "homogeneous_coordinates": homogeneous_coordinates(DEMO_FC, DEMO_CC, W, H)}


def gt_depth(depth_u16, out_wh=(320, 240)):
    """dataset.py:283-286, literally: `Image.open(depth_info).convert('F')`, `.resize((320, 240), resample=Image.NEAREST)`,
    `torch.Tensor(np.array(depth_img)) / 1000.0`, `[None, ...]` -- on an in-memory 16-bit image.  (H,W) uint16 -> (1,Ho,Wo) float32."""
    from PIL import Image
    img = Image.fromarray(np.ascontiguousarray(depth_u16).astype(np.uint16)).convert("F")
    img = img.resize(out_wh, resample=Image.NEAREST)
    return (torch.Tensor(np.array(img)) / 1000.0)[None, ...]


def nearest_table(in_size, out_size):
    """Source index per output coordinate of Pillow's NEAREST resize along one axis (third-party arithmetic, not vendored in the
    reference: src/libImaging/Geometry.c ImagingScaleAffine -- xo = a0 * 0.5, xo += a0 per pixel in double, index = xo < 0 ? -1 :
    int(xo)); pinned against Pillow itself in tests/test_preprocess.py."""
    a0 = in_size / out_size
    xo, t = a0 * 0.5, []
    for _ in range(out_size):
        t.append(-1 if xo < 0.0 else int(xo))
        xo += a0
    return np.asarray(t, np.int32)
