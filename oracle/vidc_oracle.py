"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A resolution-generic, functional PyTorch-CPU fp32 restatement of the reference's per-frame
depth-completion path (MARSLab-UMN/vi_depth_completion).  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this file; the product package
`vi_depth_completion_amd` never does and has no CPU fallback.

Parity pin: the reference has no tests of its own for this path, so this oracle is
pinned against the reference *itself*, imported in the build container through the shims in
`oracle/tools/ref_shims.py`; `oracle/tools/make_golden.py` writes the golden vectors under
`tests/golden/` and `tests/test_oracle_golden.py` checks this file against them (<= 2e-5 abs).

What each function follows (paths relative to the reference root):
  homography()            networks/warping_2dof_alignment.py:35-58   (_build_homography, _skewsymm :26-32)
  warp_geometry()         networks/warping_2dof_alignment.py:124-140 (corner bbox, kw/kh)
  warp_forward()          networks/warping_2dof_alignment.py:108-156
  warp_inverse_normals()  networks/warping_2dof_alignment.py:216-255
  resnet_pyramids()       networks/surface_normal.py:10-50, networks/depth_completion.py:16-65,
                          torchvision resnet Bottleneck (stride on conv2; un-vendored dependency, era pin 0.4.x)
  fpn_decoder()           networks/surface_normal.py:73-145, networks/depth_completion.py:75-147
  surface_normal_forward()   networks/surface_normal.py:147-171
  depth_completion_forward() networks/depth_completion.py:151-165
  mean_normal_ransac(), plane_offset_ransac(), generate_depth_from_plane(),
  extract_plane_depth()   main.py:29-190
  enrich_sparse_depth()   main.py:285-294
  call_cnn()              main.py:261-298

Third-party arithmetic under the path: ATen CPU kernels of torch 2.10 (conv2d, batch_norm,
upsample_bilinear2d(align_corners=True), grid_sampler_2d, max_pool2d).  `grid_sample` is called
like the reference does, without `align_corners`, i.e. False on torch >= 1.3;
`align_corners=True` ("as trained" on torch 1.2) is selectable.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

MAX_DEPTH_DIFF_MULTIPLIER = 10   # main.py:22
MAX_DEPTH = 10                   # main.py:23
MEAN_NORMAL_ANGLE_DIFF_THR = 20  # main.py:25


# ----------------------------------------------------------------------------------------------
# 2-DoF gravity-aligned warp
# ----------------------------------------------------------------------------------------------
class Intrinsics:
    """K, K^-1, W, H as the reference's constructor derives them (warping_2dof_alignment.py:6-24)."""

    def __init__(self, fx, fy, cx, cy):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.W = int(np.ceil(2 * cx))
        self.H = int(np.ceil(2 * cy))
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
        self.K = torch.tensor(K, dtype=torch.float32)
        self.K_inv = torch.tensor(np.linalg.inv(K), dtype=torch.float32)
        W, H = self.W, self.H
        self.corners = torch.tensor([[0, W - 1, 0, W - 1], [0, 0, H - 1, H - 1], [1, 1, 1, 1]], dtype=torch.float32)


def _skew(v):
    z = torch.zeros((), dtype=torch.float32)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def homography(g, a, intr):
    """g, a: (B,3) fp32.  Returns Cg_H_C, Cg_R_C, Cg_H_C_inv, each (B,3,3)."""
    B = g.shape[0]
    I3 = torch.eye(3, dtype=torch.float32)
    R = torch.zeros(B, 3, 3)
    for i in range(B):
        q = (-_skew(a[i])) @ g[i].view(3, 1)                     # = g x a
        dot = (a[i].view(1, 3) @ g[i].view(3, 1))[0, 0]
        nq = q.norm()
        q4 = torch.cos(0.5 * torch.atan2(nq, dot))
        # the reference's degenerate branch (:49-50) is overwritten at :53, so it is a no-op
        S = _skew((q / (2.0 * q4)).view(3))
        R[i] = I3 + 2.0 * q4 * S + 2.0 * S @ S
    Hm = intr.K @ R @ intr.K_inv
    Hinv = intr.K @ R.permute(0, 2, 1) @ intr.K_inv
    return Hm, R, Hinv


def warp_geometry(Hm, intr):
    """Per-sample (px_min, py_min, kw, kh): bbox of the 4 projected corners, 4:3-preserving scale."""
    out = []
    for i in range(Hm.shape[0]):
        c = Hm[i] @ intr.corners
        proj = c[0:2] / c[2]
        px_max, px_min = proj[0].max(), proj[0].min()
        py_max, py_min = proj[1].max(), proj[1].min()
        h_max = py_max - py_min
        w_max = px_max - px_min
        if w_max > 4 * h_max / 3:
            kw = intr.W / w_max
            kh = intr.H / (3 * w_max / 4)
        else:
            kh = intr.H / h_max
            kw = intr.W / (4 * h_max / 3)
        out.append((px_min, py_min, kw, kh))
    return out


def _pixel_grid(intr):
    ys, xs = torch.meshgrid(torch.arange(intr.H, dtype=torch.float32), torch.arange(intr.W, dtype=torch.float32), indexing="ij")
    return xs, ys


def _to_sampler_grid(u, v, intr):
    gx = 1.0 / (intr.W / 2) * (u - intr.cx)
    gy = 1.0 / (intr.H / 2) * (v - intr.cy)
    return torch.stack([gx, gy], dim=-1)


def forward_grid(g, a, intr):
    Hm, R, Hinv = homography(g, a, intr)
    geo = warp_geometry(Hm, intr)
    xs, ys = _pixel_grid(intr)
    grids = []
    for i, (px_min, py_min, kw, kh) in enumerate(geo):
        X = 1.0 / kw * xs + px_min
        Y = 1.0 / kh * ys + py_min
        P = Hinv[i] @ torch.stack([X.reshape(-1), Y.reshape(-1), torch.ones(intr.H * intr.W)])
        u = (P[0] / P[2]).view(intr.H, intr.W)
        v = (P[1] / P[2]).view(intr.H, intr.W)
        grids.append(_to_sampler_grid(u, v, intr))
    return Hm, R, torch.stack(grids)


def inverse_grid(g, a, intr):
    Hm, R, _ = homography(g, a, intr)
    geo = warp_geometry(Hm, intr)
    xs, ys = _pixel_grid(intr)
    grids = []
    for i, (px_min, py_min, kw, kh) in enumerate(geo):
        P = Hm[i] @ torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(intr.H * intr.W)])
        u = kw * (P[0] / P[2] - px_min)
        v = kh * (P[1] / P[2] - py_min)
        grids.append(_to_sampler_grid(u.view(intr.H, intr.W), v.view(intr.H, intr.W), intr))
    return Hm, R, torch.stack(grids)


def warp_forward(x, g, a, intr, align_corners=False):
    Hm, _, grid = forward_grid(g, a, intr)
    y = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    return Hm, y


def warp_inverse_normals(x, g, a, intr, align_corners=False):
    """Inverse warp of the 3-channel normal map followed by the per-pixel rotation z = R^T y."""
    Hm, R, grid = inverse_grid(g, a, intr)
    y = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    B, C, Hh, Ww = x.shape
    z = R.permute(0, 2, 1).bmm(y.view(B, C, Hh * Ww)).view(B, C, Hh, Ww)
    return Hm, z


# ----------------------------------------------------------------------------------------------
# Networks (functional over a state_dict)
# ----------------------------------------------------------------------------------------------
BN_TRAINING = False      # oracle/train_oracle.py flips this for the training step: nn.BatchNorm2d in train() mode (batch statistics,
#                          running statistics updated in place with momentum 0.1), the state the reference trains in (network_run.py:232)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=BN_TRAINING, momentum=0.1, eps=1e-5)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _bottleneck(x, sd, p, stride):
    t = F.relu(_bn(_conv(x, sd, p + "conv1"), sd, p + "bn1"))
    t = F.relu(_bn(_conv(t, sd, p + "conv2", stride, 1), sd, p + "bn2"))
    t = _bn(_conv(t, sd, p + "conv3"), sd, p + "bn3")
    if (p + "downsample.0.weight") in sd:
        x = _bn(_conv(x, sd, p + "downsample.0", stride), sd, p + "downsample.1")
    return F.relu(t + x)


def resnet_pyramids(x, sd, p, blocks=(3, 4, 23, 3)):
    c = p + "conv1."
    x = F.relu(_conv(x, sd, c + "conv1_1", 2, 1))
    x = F.relu(_bn(_conv(x, sd, c + "conv1_2", 1, 1), sd, c + "bn_2"))
    x = F.relu(_bn(_conv(x, sd, c + "conv1_3", 1, 1), sd, c + "bn1_3"))
    x = F.relu(_bn(x, sd, p + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, n in enumerate(blocks, start=1):
        for bi in range(n):
            x = _bottleneck(x, sd, "%slayer%d.%d." % (p, li, bi), 2 if (li > 1 and bi == 0) else 1)
        outs.append(x)
    return outs


def _up(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


# branch programs: 'c1'/'c3' = conv(k)+BN+ReLU occupying 3 Sequential slots, 'u<l>' = upsample to level l (1 slot)
_BRANCHES = {
    1: ["c3", "c1"],
    2: ["c1", "c3", "u1", "c1"],
    3: ["c1", "c3", "u2", "c1", "c3", "u1", "c1"],
    4: ["c1", "c3", "u3", "c1", "c3", "u2", "c1", "c3", "u1", "c1"],
}


def fpn_decoder(levels, sd, p, final_size, head_pad, final_relu, taps=None, sum_mask=None):
    """levels: [x1..x4].  Returns the full-resolution head output.  sum_mask: multiplied into z1+z2+z3+z4 (use_mask branch)."""
    sizes = {i + 1: tuple(t.shape[-2:]) for i, t in enumerate(levels)}
    zsum = None
    for b in (1, 2, 3, 4):
        t = levels[b - 1]
        idx = 0
        for op in _BRANCHES[b]:
            q = "%sfeature%d_upsamping.%d" % (p, b, idx)
            if op[0] == "c":
                k = int(op[1])
                t = F.relu(_bn(_conv(t, sd, q, 1, k // 2), sd, "%sfeature%d_upsamping.%d" % (p, b, idx + 1)))
                idx += 3
            else:
                t = _up(t, sizes[int(op[1])])
                idx += 1
        if taps is not None:
            taps["z%d" % b] = t
        zsum = t if zsum is None else zsum + t
    if sum_mask is not None:
        zsum = zsum * sum_mask
    h = F.relu(_conv(zsum, sd, p + "feature_concat.0", 1, 1))
    h = _conv(h, sd, p + "feature_concat.2", 1, head_pad)
    if taps is not None:
        taps["head_lowres"] = h
    h = _up(h, final_size)
    return F.relu(h) if final_relu else h


def surface_normal_forward(sd, x, g, a, intr, align_corners=False, taps=None, use_mask=False):
    _, xw = warp_forward(x, g, a, intr, align_corners)
    levels = resnet_pyramids(xw, sd, "resnet_pyramids.")
    sum_mask = None
    if use_mask:          # networks/surface_normal.py:150-162
        fm = (xw[:, 0:1] + xw[:, 1:2] + xw[:, 2:3] > 1e-2).float()
        masks = [F.interpolate(fm, size=tuple(t.shape[-2:]), mode="nearest") for t in levels]
        levels = [t * m for t, m in zip(levels, masks)]
        sum_mask = masks[0]
    y = fpn_decoder(levels, sd, "", (intr.H, intr.W), 0, False, taps, sum_mask=sum_mask)
    _, z = warp_inverse_normals(y, g, a, intr, align_corners)
    if taps is not None:
        taps.update(warped=xw, x1=levels[0], x2=levels[1], x3=levels[2], x4=levels[3], normal_raw=y)
    return F.normalize(z, dim=1)


def depth_completion_forward(sd, image, normal, depth, taps=None):
    li = resnet_pyramids(image, sd, "resnet_rgb.")
    ln = resnet_pyramids(normal, sd, "resnet_normal.")
    ld = resnet_pyramids(depth, sd, "resnet_depth.")
    levels = [torch.cat(t, dim=1) for t in zip(li, ln, ld)]
    return fpn_decoder(levels, sd, "", tuple(image.shape[-2:]), 1, True, taps)


# ----------------------------------------------------------------------------------------------
# Plane block (normal -> plane -> depth projection) and sparse-depth enrichment
# ----------------------------------------------------------------------------------------------
def dorn_forward(sd, x, taps=None):
    """SurfaceNormalDORN.forward (networks/surface_normal_dorn.py:151-154): ResNet (:130-141, layer3/4 at stride 1, :119-125) ->
    SceneUnderstandingModuleBN (:79-89) with FullImageEncoder (:18-30) -> F.normalize.  Eval mode: Dropout2d is the identity."""
    p = "feature_extractor."
    t = F.relu(_conv(x, sd, p + "conv1.conv1_1", 2, 1))
    t = F.relu(_bn(_conv(t, sd, p + "conv1.conv1_2", 1, 1), sd, p + "conv1.bn_2"))
    t = F.relu(_bn(_conv(t, sd, p + "conv1.conv1_3", 1, 1), sd, p + "conv1.bn1_3"))
    t = F.relu(_bn(t, sd, p + "bn1"))
    t = F.max_pool2d(t, 3, 2, 1)
    for li, (n, stride) in enumerate(zip((3, 4, 23, 3), (1, 2, 1, 1)), start=1):
        for bi in range(n):
            t = _bottleneck(t, sd, "%slayer%d.%d." % (p, li, bi), stride if bi == 0 else 1)
    if taps is not None:
        taps["features"] = t
    a = "aspp_module."
    e = F.avg_pool2d(t, 8, stride=8, padding=(1, 0))
    e = F.relu(F.linear(e.reshape(-1, 2048 * 4 * 5), sd[a + "encoder.global_fc.weight"], sd[a + "encoder.global_fc.bias"]))
    e = F.conv2d(e.view(-1, 512, 1, 1), sd[a + "encoder.conv1.weight"], sd[a + "encoder.conv1.bias"])
    x1 = F.interpolate(e, size=(t.shape[2], t.shape[3]), mode="bilinear", align_corners=True)
    outs = [x1]
    for name, dil in (("aspp1", 0), ("aspp2", 6), ("aspp3", 12), ("aspp4", 18)):
        w, b = sd[a + name + ".0.weight"], sd[a + name + ".0.bias"]
        u = F.conv2d(t, w, b) if dil == 0 else F.conv2d(t, w, b, padding=dil, dilation=dil)
        u = F.relu(_bn(u, sd, a + name + ".1"))
        u = F.relu(_bn(F.conv2d(u, sd[a + name + ".3.weight"], sd[a + name + ".3.bias"]), sd, a + name + ".4"))
        outs.append(u)
    c = torch.cat(outs, dim=1)
    if taps is not None:
        taps["concat"] = c
    h = F.relu(F.conv2d(c, sd[a + "concat_process.1.weight"], sd[a + "concat_process.1.bias"]))
    y = F.conv2d(h, sd[a + "concat_process.4.weight"], sd[a + "concat_process.4.bias"])
    y = F.interpolate(y, size=(x.shape[2], x.shape[3]), mode="bilinear", align_corners=True)
    return F.normalize(y, dim=1)


def mean_normal(n):
    return F.normalize(n.mean(dim=0), dim=0)


def mean_normal_ransac(normals, angle_thr=20.0, num_hyp=300, rng=np.random, info=None):
    """main.py:38-62.  `info` (dict, tests only) receives the row of the winning hypothesis."""
    N = normals.shape[0]
    idx = rng.permutation(np.r_[0:N])[0:min(num_hyp, N)]
    dots = torch.clamp(normals[idx] @ normals.t(), -1.0, 1.0)
    close = torch.acos(dots) * (180.0 / np.pi) < angle_thr
    win = torch.argmax(close.sum(dim=1)).item()
    best = close[win]
    if info is not None:
        info["winner_normal"] = normals[idx[win]].clone()
    inl = normals[best]
    m = mean_normal(inl)
    ang = torch.acos(torch.clamp(inl @ m[:, None], -1, 1)) * (180 / np.pi)
    return m, ang, best, idx


def plane_offset_ransac(normal, pts, dist_thr=1.0e-1, min_inliers=1, num_hyp=300, rng=np.random):
    M = pts.shape[1]
    if M == 0:
        return 0, 0
    idx = np.r_[0:M] if M <= num_hyp else rng.permutation(np.r_[0:M])[0:num_hyp]
    dots = (normal[None, :] @ pts).squeeze()
    if dots.nelement() == 1:
        return (-dots, 1) if min_inliers == 1 else (0, 0)
    hyp = -dots[idx]
    inl = torch.abs(hyp[..., None] + dots[None, ...]) < dist_thr
    cnt = inl.sum(dim=1)
    bi = torch.argmax(cnt).item()
    if cnt[bi] < min_inliers:
        return 0, 0
    return -torch.mean(dots[inl[bi]]), int(inl[bi].sum().item())


def generate_depth_from_plane(plane_eq, mask, homo, depth_image, mean_depth):
    dots = torch.sum(homo * plane_eq[0:3][None, None, :], dim=2)
    mask2 = mask & (dots.abs() > 1e-3)
    vals = (-plane_eq[3] / dots)[mask2]
    n = vals.nelement()
    if (torch.sum(vals > mean_depth * MAX_DEPTH_DIFF_MULTIPLIER) / n > 0.05) or (torch.sum(vals > MAX_DEPTH) / n > 0):
        return False
    if torch.sum(vals < 0) / n > 0.0:
        return False
    depth_image[mask2] = vals
    return True


def extract_plane_depth(normal_image, mask, depth, homo, rng=np.random, trace=None):
    """normal_image (3,H,W), mask (H,W) integer ids, depth (H,W), homo (H,W,3) -> planes_depth (H,W).
    (The reference also returns a plane-normal image which its only caller discards, main.py:282.)"""
    classes = torch.unique(mask)
    if int(classes.max()) + 1 == 1:
        return depth
    nimg = normal_image.permute(1, 2, 0)
    out = depth.clone()
    for cls in classes:
        if cls == 0:
            continue
        m = mask == cls
        normals = nimg[m]
        winfo = {}
        n_bar, ang, best, idx = mean_normal_ransac(normals, rng=rng, info=winfo)
        m2 = m.clone()
        m2[m] = best
        rec = {"cls": int(cls), "winner_normal": winfo["winner_normal"], "hyp_idx": np.asarray(idx).copy(), "n_bar": n_bar.clone(), "n_inl": int(best.sum()),
               "mean_angle": float(ang.abs().mean()), "accepted": False, "offset": 0.0, "n_off_inl": 0, "valid": False}
        if trace is not None:
            trace.append(rec)
        if ang.abs().mean() > MEAN_NORMAL_ANGLE_DIFF_THR:
            continue
        rec["accepted"] = True
        vd = m2 & (depth > 0)
        pc = homo[vd] * depth[vd][:, None]
        mean_depth = torch.mean(depth[vd])
        off, n_inl = plane_offset_ransac(n_bar, pc.T, rng=rng)
        rec["n_off_inl"] = int(n_inl)
        if n_inl == 0:
            continue
        rec["offset"] = float(off)
        eq = torch.zeros(4)
        eq[0:3] = n_bar
        eq[3] = off
        rec["valid"] = bool(generate_depth_from_plane(eq, m2, homo, out, mean_depth))
    keep = depth > 0
    out[keep] = depth[keep]
    return out


def enrich_sparse_depth(ds, di, goal, rng=np.random, trace=None):
    """ds, di: (B,1,H,W).  Copies <= goal randomly drawn plane-depth pixels into a clone of ds."""
    out = ds.clone()
    for b in range(ds.shape[0]):
        nz = torch.nonzero(di[b, 0] > 0, as_tuple=True)
        n = len(nz[0])
        k = min(goal, n)
        sub = np.unique(rng.randint(0, n, size=k)) if n > 0 else np.zeros(0, dtype=np.int64)
        if trace is not None:
            trace.append({"nnz": n, "sub": sub.copy()})
        out[b, 0, nz[0][sub], nz[1][sub]] = di[b, 0, nz[0][sub], nz[1][sub]]
    return out


def call_cnn(sn_sd, dc_sd, batch, plane_masks, intr, enriched_samples=200, align_corners=False, rng=np.random,
             taps=None):
    """The whole per-batch hot path, main.py:261-298.  plane_masks: list of (H,W) integer id arrays."""
    ds, rgb = batch["sparse_depth"], batch["image"]
    normals = surface_normal_forward(sn_sd, rgb, batch["gravity"], batch["aligned_direction"], intr, align_corners)
    if taps is not None:
        taps["normals"] = normals
    if enriched_samples == 0:
        return depth_completion_forward(dc_sd, rgb, normals, ds)
    homo = batch["homogeneous_coordinates"]
    di = torch.zeros_like(ds)
    ptrace, etrace = [], []
    for i in range(ds.shape[0]):
        pm = torch.as_tensor(np.asarray(plane_masks[i])).view(ds.shape[-2], ds.shape[-1])
        di[i, 0] = extract_plane_depth(normals[i], pm, ds[i, 0], homo[i], rng=rng, trace=ptrace)
    enriched = enrich_sparse_depth(ds, di, enriched_samples, rng=rng, trace=etrace)
    if taps is not None:
        taps.update(plane_depth=di, enriched=enriched, plane_trace=ptrace, enrich_trace=etrace)
    return depth_completion_forward(dc_sd, rgb, normals, enriched)
