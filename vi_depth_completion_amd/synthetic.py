"""Seeded synthetic weights and inputs.

No checkpoints ship with the reference (README.md:88-90) and there is no network, so parity
and the bench run on a *calibrated seeded state_dict*: every tensor is a pure function of
(seed, parameter name, element index) computed with 64-bit integer hashing plus one float64
multiply -- no transcendental functions, so the same bits come out on any host, on CPU or on
the GPU, and the 1.5 GB of weights never have to be committed.

The recipe keeps activations O(1) through the 33-block residual stacks (small gain on the
last BatchNorm of every bottleneck) and gives a non-degenerate, metre-scale depth output
(default init gives an all-zero depth map).
"""
import zlib

import numpy as np
import torch

_M64 = (1 << 64) - 1
_GOLD = 0x9E3779B97F4A7C15


def _to_i64(v):
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


_C1 = _to_i64(0xBF58476D1CE4E5B9)
_C2 = _to_i64(0x94D049BB133111EB)


def _lsr(x, n):
    # logical shift right on int64 tensors (torch's >> is arithmetic)
    return (x >> n) & ((1 << (64 - n)) - 1)


def _mix64(x):
    """splitmix64 finaliser on an int64 tensor (two's-complement wraparound == mod 2^64)."""
    x = (x ^ _lsr(x, 30)) * _C1
    x = (x ^ _lsr(x, 27)) * _C2
    return x ^ _lsr(x, 31)


def _stream_base(seed, name):
    h = (zlib.crc32(name.encode()) & 0xFFFFFFFF) | ((zlib.adler32(name.encode()) & 0xFFFFFFFF) << 32)
    return (seed * 0xD1342543DE82EF95 + h * _GOLD + 0x2545F4914F6CDD1D) & _M64


_CHUNK = 1 << 20


def _hash_into(seed, name, n, device, finish, out):
    """Hash element indices 0..n-1 of stream (seed, name) chunk-wise (cache resident, in place) and
    write finish(hash_chunk) into out[chunk]."""
    base = _to_i64(_stream_base(seed, name))
    gold = _to_i64(_GOLD)
    step = n if device != "cpu" else _CHUNK
    for s0 in range(0, n, step):
        s1 = min(n, s0 + step)
        x = torch.arange(s0, s1, dtype=torch.int64, device=device)
        x.mul_(gold).add_(base)
        for sh, mul in ((30, _C1), (27, _C2), (31, None)):
            t = x >> sh
            t.bitwise_and_((1 << (64 - sh)) - 1)
            x.bitwise_xor_(t)
            if mul is not None:
                x.mul_(mul)
        out[s0:s1] = finish(x)
    return out


def _numel(shape):
    return int(np.prod(shape)) if len(shape) else 1


def uniform01(seed, name, shape, device="cpu"):
    """U[0,1) with 24-bit resolution, float32, exactly reproducible."""
    n = _numel(shape)
    out = torch.empty(n, dtype=torch.float32, device=device)

    def fin(h):
        return _lsr(h, 40).to(torch.float32).mul_(1.0 / (1 << 24))   # 24-bit ints are exact in fp32

    return _hash_into(seed, name, n, device, fin, out).reshape(shape)


def normal01(seed, name, shape, device="cpu", scale=1.0, dtype=torch.float64):
    """Approximately N(0,scale^2): Irwin-Hall sum of four 16-bit uniforms, one float64 multiply."""
    n = _numel(shape)
    out = torch.empty(n, dtype=dtype, device=device)
    k = float(scale) / 37837.22   # sum of 4 U{0..65535}: mean 2*65535, std 37837.22

    def fin(h):
        s = (h & 0xFFFF) + (_lsr(h, 16) & 0xFFFF) + (_lsr(h, 32) & 0xFFFF) + _lsr(h, 48)
        return s.sub_(2 * 65535).to(torch.float64).mul_(k).to(dtype)

    return _hash_into(seed, name, n, device, fin, out).reshape(shape)


def seeded_state_dict(reference_state, seed=1234, device="cpu"):
    """Fill a state_dict-shaped mapping (name -> tensor, only shapes/dtypes are read).

    conv weight : kaiming-normal (fan_in, gain sqrt(2))
    conv bias   : 0.05*N(0,1); the 1-channel depth head (`feature_concat.2`, Cout=1) gets weight*0.25, bias +2.0
    BN weight   : U(0.5,1.5), except the last BN of a bottleneck (`.bn3.`) U(0.04,0.12)
    BN bias     : 0.1*N(0,1);  running_mean 0.1*N(0,1);  running_var U(0.5,1.5)
    """
    out = {}
    for name, ref in reference_state.items():
        shape = tuple(ref.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=torch.int64, device=device)
            continue
        leaf = name.rsplit(".", 1)[-1]
        if len(shape) == 4:  # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.25 if (name.endswith("feature_concat.2.weight") and shape[0] == 1) else 1.0
            out[name] = normal01(seed, name, shape, device, scale=gain * float(np.sqrt(2.0 / fan_in)), dtype=torch.float32)
        elif len(shape) == 2 and leaf == "weight":  # nn.Linear weight: kaiming-normal like a conv
            out[name] = normal01(seed, name, shape, device, scale=float(np.sqrt(2.0 / shape[1])), dtype=torch.float32)
        elif leaf == "running_var":
            out[name] = uniform01(seed, name, shape, device) + 0.5
        elif leaf == "running_mean":
            out[name] = (normal01(seed, name, shape, device) * 0.1).to(torch.float32)
        elif leaf == "weight":  # BN gamma
            u = uniform01(seed, name, shape, device)
            out[name] = (u * 0.08 + 0.04) if ".bn3." in name else (u + 0.5)
        elif leaf == "bias":
            b = normal01(seed, name, shape, device)
            if name.endswith("feature_concat.2.bias") and shape == (1,):
                out[name] = torch.full(shape, 2.0, dtype=torch.float32, device=device)
            elif _is_bn_bias(name, reference_state):
                out[name] = (b * 0.1).to(torch.float32)
            else:
                out[name] = (b * 0.05).to(torch.float32)
        else:
            raise KeyError("unexpected state_dict entry " + name)
    return out


def _is_bn_bias(name, state):
    return (name[: -len("bias")] + "running_var") in state


def plane_id_map(height, width):
    """Fixed plane-instance id map of config C2 ("plane mask fixed"):
    rows >= H/2 -> id 1; rows < H/2 and cols < W/2 -> id 2; else 0."""
    m = np.zeros((height, width), dtype=np.uint8)
    m[height // 2:, :] = 1
    m[: height // 2, : width // 2] = 2
    return m


def homogeneous_grid(fc, cc, width, height):
    """Per-pixel ((x-cx)/fx, (y-cy)/fy, 1), float32 (H,W,3); follows dataset.py:34-42."""
    xx, yy = np.meshgrid(np.arange(width), np.arange(height))
    h = np.ones((height, width, 3), dtype=np.float64)
    h[:, :, 0] = (xx - cc[0]) / fc[0]
    h[:, :, 1] = (yy - cc[1]) / fc[1]
    return torch.from_numpy(h.astype(np.float32))


DEMO_FC = (202.9953, 202.9540)   # dataset.py:456-457
DEMO_CC = (159.7645, 122.0951)


def synthetic_batch(batch, height=240, width=320, seed=1234, n_sparse=200, frame0=0):
    """C2 synthetic input batch, keyed like DemoDataset.__getitem__ (dataset.py:515-520).
    Frame f of any batch is a function of (seed, frame0+f) only, so shards are world-size independent."""
    imgs, gs, als, sds = [], [], [], []
    for f in range(frame0, frame0 + batch):
        tag = "f%d" % f
        imgs.append(uniform01(seed, tag + ".image", (3, height, width)))
        n = normal01(seed, tag + ".g", (2,))
        g = torch.tensor([0.08 * float(n[0]), 1.0, 0.12 * float(n[1])], dtype=torch.float64)
        gs.append((g / g.norm()).to(torch.float32))
        als.append(torch.tensor([0.0, 1.0, 0.0]))
        sd = torch.zeros(height * width, dtype=torch.float32)
        pos = (uniform01(seed, tag + ".pos", (n_sparse,)).double() * (height * width)).long().clamp_(max=height * width - 1)
        dep = uniform01(seed, tag + ".dep", (n_sparse,)) * 4.5 + 0.5
        sd[pos] = dep
        sds.append(sd.view(1, height, width))
    cc = (DEMO_CC[0], DEMO_CC[1] * height / 240.0)
    homo = homogeneous_grid(DEMO_FC, cc, width, height)
    return {
        "image": torch.stack(imgs),
        "sparse_depth": torch.stack(sds),
        "gravity": torch.stack(gs),
        "aligned_direction": torch.stack(als),
        "homogeneous_coordinates": homo.unsqueeze(0).repeat(batch, 1, 1, 1),
        "color_filename": ["color_%06d.png" % f for f in range(frame0, frame0 + batch)],
    }


def synthetic_ground_truth_depth(image, seed=1234):
    """Dense synthetic ground-truth depth (metres) for the training configuration (BASELINE configs[4]): a smooth function of the
    frame content plus noise, ~20 % invalid (0) pixels like a real sensor map.  image: (B,3,H,W) -> (B,1,H,W)."""
    B, _, H, W = image.shape
    base = 1.0 + 3.0 * image.mean(dim=1, keepdim=True) + 0.5 * uniform01(seed, "gt.noise", (B, 1, H, W))
    hole = uniform01(seed, "gt.hole", (B, 1, H, W)) < 0.2
    return torch.where(hole, torch.zeros_like(base), base).float()


def synthetic_camera_batch(batch, src_height=480, src_width=640, seed=1234, n_sparse=200, frame0=0, out_hw=(240, 320)):
    """Raw camera-side inputs of a stream (BASELINE configs[2], [3]: "640x480 stream", "1280x720"), i.e. what DemoDataset reads from
    disk before its own pre-processing (dataset.py:461-510): uint8 RGB frames (B,H,W,3), the raw gravity file values (B,3) (the loader
    negates y and z) and per frame a (n,4) float64 array of VI-SLAM tracks (id, X, Y, Z) whose projections fall inside the
    out_hw image.  Frame f is a function of (seed, frame0+f) only."""
    imgs, gravs, tracks = [], [], []
    Ho, Wo = out_hw
    for f in range(frame0, frame0 + batch):
        tag = "cam%d" % f
        imgs.append((uniform01(seed, tag + ".image", (src_height, src_width, 3)) * 256.0).clamp_(max=255.0).to(torch.uint8))
        n = normal01(seed, tag + ".g", (2,))
        g = torch.tensor([0.08 * float(n[0]), 1.0, 0.12 * float(n[1])], dtype=torch.float64)
        g = (g / g.norm()).numpy()
        gravs.append([g[0], -g[1], -g[2]])
        u = uniform01(seed, tag + ".uv", (n_sparse, 2)).double().numpy()
        z = (uniform01(seed, tag + ".z", (n_sparse,)).double() * 4.5 + 0.5).numpy()
        col, row = u[:, 0] * (Wo - 1) + 0.5, u[:, 1] * (Ho - 1) + 0.5
        x = (col - DEMO_CC[0]) / DEMO_FC[0] * z
        y = (row - DEMO_CC[1] * Ho / 240.0) / DEMO_FC[1] * z
        tracks.append(np.stack([np.arange(n_sparse, dtype=np.float64), x, y, z], axis=1))
    return {"image_u8": torch.stack(imgs), "gravity_raw": np.asarray(gravs, dtype=np.float64), "klt_tracks": tracks}


def seeded_detector_state_dict(reference_state, seed, device="cpu"):
    """Seeded parameters for the plane-mask detector (networks/plane_mask_rcnn.py; the reference's GeneralizedRCNN state_dict layout,
    SURVEY.md §8f-1): `seeded_state_dict` for every learnable / FrozenBatchNorm entry, the anchor buffers kept as they are, and four
    calibrations so that random weights give a non-degenerate detector: the stem's BN gain divides the 0..255-scale input down to O(1)
    activations (otherwise every sigmoid saturates and top-k / NMS decisions are ties), the class head is biased towards the plane
    class and the mask head towards "inside", so that a few detections pass the 0.9 confidence threshold with large masks."""
    learn = {k: v for k, v in reference_state.items() if "anchor_generator" not in k}
    out = seeded_state_dict(learn, seed, device)
    for k, v in reference_state.items():
        if "anchor_generator" in k:
            out[k] = v.detach().clone().to(device)
    out["backbone.body.stem.bn1.weight"] = out["backbone.body.stem.bn1.weight"] / 70.0
    for k in list(out):          # the residual stages still grow the activations ~20x: the FPN laterals bring the pyramid back to O(1)
        if ".fpn.fpn_inner" in k:
            out[k] = out[k] * 0.05
    if "roi_heads.box.predictor.cls_score.bias" in out:
        out["roi_heads.box.predictor.cls_score.bias"] = torch.tensor([-1.0, 1.0], dtype=torch.float32, device=device)
    if "roi_heads.mask.predictor.mask_fcn_logits.bias" in out:
        out["roi_heads.mask.predictor.mask_fcn_logits.bias"] = torch.tensor([0.0, 0.75], dtype=torch.float32, device=device)
    return out
