"""Host side of the HIP engine: records a network as a `Program` (array of C launch descriptors over
device buffers), plans buffers, folds BatchNorm, packs weights, and runs the whole thing through ONE
C-ABI call per frame (`vidc_program_run` / hipGraph replay).  PyTorch is used for device memory and
streams only; there is no eager or CPU execution path here.

Layout: activations are NHWC fp32.  A "grouped" tensor (G > 1) keeps the G groups as channel slices of
one [B,H,W,G*C] buffer, so the rgb/normal/depth pyramids of ModifiedFPN run as one grouped launch per
layer and `torch.cat((rgb, normal, depth), dim=1)` (depth_completion.py:151-152) costs nothing.
"""
import ctypes as C
import json
import os

import torch

from . import _lib as L


class K(tuple):
    """Tuple of parameter-name prefixes (one per group); `K + 'conv1'` appends to every member."""

    def __new__(cls, items):
        return super().__new__(cls, (items,) if isinstance(items, str) else tuple(items))

    def __add__(self, s):
        return K(p + s for p in self)


def _keys(k):
    return k if isinstance(k, K) else K(k)


class T:
    """Program tensor: NHWC view (channel slice) of a planned buffer; NCHW for program inputs/outputs."""
    __slots__ = ("buf", "B", "H", "W", "C", "G", "ld", "ch_off", "nchw")

    def __init__(self, buf, B, H, W, Cc, G=1, ld=None, ch_off=0, nchw=False):
        self.buf, self.B, self.H, self.W, self.C, self.G = buf, B, H, W, Cc, G
        self.ld = ld if ld is not None else G * Cc
        self.ch_off, self.nchw = ch_off, nchw

    @property
    def numel(self):
        return self.B * self.H * self.W * self.ld


class WeightStore:
    """Device-resident derived parameters of one nn.Module: packed conv weights and folded BN affines.
    Rebuilt lazily after `invalidate()` (load_state_dict / .cuda() / .to())."""

    def __init__(self, module):
        self.module = module
        self._cache = {}
        self._sd = None
        self._virtual = {}       # name -> fn(state_dict) : derived entries (e.g. a Linear weight re-laid-out for NHWC activations)
        self.bn_eps = 1e-5       # nn.BatchNorm2d's default; 0 for the detector's FrozenBatchNorm2d (layers/batch_norm.py: no eps)

    def add_virtual(self, name, fn):
        if name not in self._virtual:
            self._virtual[name] = fn
            self._sd = None

    def invalidate(self):
        self._cache.clear()
        self._sd = None

    def sd(self):
        if self._sd is None:
            self._sd = {k: v.detach() for k, v in self.module.state_dict().items()}
            for name, fn in self._virtual.items():
                self._sd[name] = fn(self._sd)
        return self._sd

    def raw(self, key):
        return self.sd()[key]

    def packed(self, keys, dry_run=False, precision=0):
        """[G][Cout][KH][KW][Cin] fp32 (precision 0) or the split-bf16 image of the same bytes (precision 1), packed by
        the C kernels."""
        ck = ("w", precision) + tuple(keys)
        if ck not in self._cache:
            ws = [self.sd()[k + ".weight"].contiguous() for k in keys]
            co, ci, kh, kw = ws[0].shape
            out = torch.empty((len(ws), co, kh * kw * ci), dtype=torch.float32, device=ws[0].device)
            for g, w in enumerate(ws):
                assert tuple(w.shape) == (co, ci, kh, kw)
                if not dry_run:
                    fn = L.lib().vidc_pack_conv_weight_bf16x3 if precision == L.PREC_BF16X3 else L.lib().vidc_pack_conv_weight
                    L.check(fn(L.ptr(w), L.ptr(out[g]), co, ci, kh, kw, L.current_stream()), "pack")
            if dry_run:
                return out
            self._cache[ck] = out
        return self._cache[ck]

    def packed_winograd(self, keys, m, dry_run=False, precision=0):
        """U = G g G^T of the 3x3 filters `keys` for F(m x m, 3x3): [G][(m+2)^2][Cout][Cin] fp32 (csrc/winograd.hip, fp64 inside), or the
        split-bf16 image of the same bytes (precision 1) -- the weights of the (m+2)^2 * G group 1x1 GEMM launch."""
        ck = ("wino", m, precision) + tuple(keys)
        if ck not in self._cache:
            ws = [self.sd()[k + ".weight"].contiguous() for k in keys]
            co, ci, kh, kw = ws[0].shape
            assert (kh, kw) == (3, 3)
            a2 = (m + 2) * (m + 2)
            out = torch.empty((len(ws) * a2, co, ci), dtype=torch.float32, device=ws[0].device)
            if dry_run:
                return out
            for g, w in enumerate(ws):
                assert tuple(w.shape) == (co, ci, 3, 3)
                L.check(L.lib().vidc_winograd_weight_transform(L.ptr(w), L.ptr(out[g * a2]), co, ci, m, L.current_stream()), "winograd weights")
            if precision == L.PREC_BF16X3:
                img = torch.empty_like(out)
                L.check(L.lib().vidc_pack_conv_weight_bf16x3(L.ptr(out), L.ptr(img), len(ws) * a2 * co, ci, 1, 1, L.current_stream()), "pack")
                out = img
            self._cache[ck] = out
        return self._cache[ck]

    def packed_winograd_fused(self, keys, dry_run=False):
        """U of F(4 x 4, 3x3) in the fragment order of the one-launch kernel (vidc_winograd_weight_pack_fused): 36 Cout Cin floats per group."""
        ck = ("winof",) + tuple(keys)
        if ck not in self._cache:
            u = self.packed_winograd(keys, 4, dry_run, 0)
            out = torch.empty_like(u)
            if dry_run:
                return out
            co, ci = u.shape[1], u.shape[2]
            for g in range(len(keys)):
                L.check(L.lib().vidc_winograd_weight_pack_fused(L.ptr(u[g * 36]), L.ptr(out[g * 36]), co, ci, L.current_stream()), "winograd weights (fused order)")
            self._cache[ck] = out
        return self._cache[ck]

    def identity_affine(self, co, device):
        ck = ("id", co, str(device))
        if ck not in self._cache:
            self._cache[ck] = (torch.ones(co, dtype=torch.float32, device=device), torch.zeros(co, dtype=torch.float32, device=device))
        return self._cache[ck]

    def affine(self, conv_keys, bn_keys):
        """Per-output-channel (scale, shift) [G][Cout] for conv(+bias) followed by eval-mode BN, folded in fp64."""
        ck = ("a",) + tuple(conv_keys) + tuple(bn_keys or ())
        if ck not in self._cache:
            sd = self.sd()
            scales, shifts = [], []
            for g, ckey in enumerate(conv_keys):
                co = sd[ckey + ".weight"].shape[0] if ckey is not None else sd[bn_keys[g] + ".weight"].shape[0]
                dev = sd[(ckey or bn_keys[g]) + ".weight"].device
                bias = sd.get(ckey + ".bias") if ckey is not None else None
                bias = bias.double() if bias is not None else torch.zeros(co, dtype=torch.float64, device=dev)
                if bn_keys is not None:
                    bn = bn_keys[g]
                    s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + self.bn_eps)
                    t = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * s + bias * s
                else:
                    s, t = torch.ones(co, dtype=torch.float64, device=dev), bias
                scales.append(s.float())
                shifts.append(t.float())
            self._cache[ck] = (torch.stack(scales).contiguous(), torch.stack(shifts).contiguous())
        return self._cache[ck]


class JointWeightStore(WeightStore):
    """Derived parameters of several modules behind one key space: key "name/param" resolves to `modules[name]`'s
    state_dict entry "param".  Lets one Program (one grouped launch per layer) span networks -- the software-pipelined
    frame program of pipeline.InterleavedPrograms runs the surface-normal pyramid of frame i+1 and the three
    depth-completion pyramids of frame i as four groups of the same launches."""

    def __init__(self, modules):
        super().__init__(None)
        self.modules = dict(modules)

    def sd(self):
        if self._sd is None:
            self._sd = {"%s/%s" % (name, k): v.detach() for name, m in self.modules.items() for k, v in m.state_dict().items()}
            for name, fn in self._virtual.items():
                self._sd[name] = fn(self._sd)
        return self._sd


_TUNING = None


def tuning_table():
    """Optional per-shape (tile, splitk) overrides measured on MI355X (tools/autotune.py writes it)."""
    global _TUNING
    if _TUNING is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_tuning.json")
        _TUNING = json.load(open(path)) if os.path.exists(path) else {}
        if os.environ.get("VIDC_TUNING_OVERRIDE"):          # experiment knob: '{"M80_N256_K256_k1s1_G144": [40, 64]}' replaces / adds entries (A-B runs of one signature)
            _TUNING = dict(_TUNING)
            _TUNING.update(json.loads(os.environ["VIDC_TUNING_OVERRIDE"]))
    return _TUNING


def precision_mode():
    """VIDC_PRECISION=fp32  : every conv on fp32 MFMA (exact fp32; the reference mode).
       VIDC_PRECISION=mixed : (default) compute-bound convs run split-bf16 3-pass MFMA (per-shape choice from the measured
                              table, else by size); whole-path depth RMSE vs fp32 ~1.4e-5, bar 1e-3."""
    return os.environ.get("VIDC_PRECISION", "mixed")


def default_precision(flops):
    return L.PREC_BF16X3 if flops >= 1.5e9 else L.PREC_FP32


# LDS bytes of the tilings (NS * (BM + BN) * 32 floats * WKW) and, for the co-residency experiment, the nearest tiling of at most 80 KB
_TILE_LDS_KB = {33: 128, 34: 96, 35: 64, 36: 96, 37: 128, 38: 120, 39: 120, 28: 48, 29: 32, 30: 64, 31: 48, 32: 48, 1: 64, 2: 72, 3: 72, 4: 64, 5: 96, 6: 72, 7: 96, 8: 80, 9: 128, 10: 120, 11: 128, 12: 120, 13: 128, 14: 72, 15: 120, 16: 128, 17: 64,
                18: 128, 19: 72, 20: 72, 21: 72, 22: 120, 23: 120, 24: 96, 25: 96, 26: 144, 27: 144}
_TILE_SMALL = {5: 4, 13: 4, 18: 4, 7: 6, 9: 6, 10: 6, 11: 6, 15: 6, 16: 6, 12: 8, 22: 21, 23: 21, 24: 1, 25: 1, 26: 1, 27: 1}


def _capped_tile(tile, cap_kb):
    return _TILE_SMALL.get(tile, tile) if _TILE_LDS_KB.get(tile, 0) > cap_kb else tile


def winograd_mode():
    """VIDC_WINOGRAD=auto : (default) 3x3 / stride 1 / pad 1 convs with >= 128 input channels run as Winograd F(m x m, 3x3) GEMMs
                             (csrc/winograd.hip): the entry "W:<direct signature>" of the measured table picks m in {0, 2, 4} (5 = m 4 in ONE launch, fp32 only), else (fp32 mode
                             only) m = 4 on maps of at least 24 rows and columns and m = 2 below; the mixed mode without a table entry stays direct.
       VIDC_WINOGRAD=0 / 2 / 4 : never / always that m where the layer qualifies (tests, A/B runs)."""
    return os.environ.get("VIDC_WINOGRAD", "auto")


def winograd_choice(B, H, W, co, ci, kh, kw, stride, padding, dilation, G, mode=None, precision=None):
    if (kh, kw, stride, padding, dilation) != (3, 3, 1, 1, 1) or ci % 32 or co % 32:
        return 0
    mode = mode if mode is not None else winograd_mode()
    precision = precision if precision is not None else precision_mode()
    if mode in ("0", "2", "4"):
        m = int(mode)
    else:
        ent = None                                # [m in the fp32 mode, m in the mixed mode]; measured on the frame program (4 pyramid groups):
        for gg in (G, 4, 3, 1, 2):                # the stand-alone programs run the same layers with 1 / 3 groups and take that verdict
            ent = tuning_table().get("W:M%d_N%d_K%d_k3s1_G%d" % (B * H * W, co, 9 * ci, gg))
            if ent is not None:
                break
        if ent is not None:
            m = int(ent[1 if precision == "mixed" else 0]) if isinstance(ent, (list, tuple)) else int(ent)
            if m == 5 and gg != G:                 # 5 = F(4 x 4) in ONE launch (csrc/wfused.hip): measured for exactly that group count
                m = 4
        elif ci < 128 or precision == "mixed":
            # no measured verdict for this shape.  fp32: the heuristic below.  mixed: the direct form -- in the committed table the bf16x3
            # direct conv beats the transform + GEMM + transform triple for 73 of 112 layers (three launches and two HBM round trips through
            # V and M against a conv that is cheap already), so an untuned resolution or batch must not default to the slower form
            m = 0
        else:
            m = 4 if min(H, W) >= 24 else 2
    if m:
        mt = 4 if m == 5 else m
        a2 = (mt + 2) * (mt + 2)
        tiles = B * (-(-H // mt)) * (-(-W // mt))
        if tiles * a2 * G * max(ci, co) * 4 >= 2 ** 31 or B * (-(-H // mt)) > 65535:     # 32-bit buffer offsets in the GEMM kernel / grid.y
            return 0
    return m


def conv_signature(d):
    return "M%d_N%d_K%d_k%ds%d_G%d" % (d.B * d.Ho * d.Wo, d.Cout, d.KH * d.KW * d.Cin, d.KH, d.stride, d.groups)


class Program:
    """Records ops symbolically (buffers are integers), then `finalize()` plans memory and builds the C program."""

    def __init__(self, weights, device, batch, reuse_buffers=True, mode=None, winograd=None):
        """mode: "fp32" | "mixed" (default: precision_mode()); winograd: "0" | "2" | "4" | "auto" (default: winograd_mode()) -- a program whose
        results feed DECISIONS (the plane-mask detector: score / IoU / mask thresholds) is recorded with mode="fp32", winograd="0": exact
        fp32 direct-form sums, the arithmetic its oracle pins."""
        self.ws, self.device, self.B = weights, device, batch
        self.ops = []            # (kind, dict)
        self.buf_elems = []      # elements per buffer id
        self.pinned = set()      # buffer ids that must not alias (inputs / outputs)
        self.inputs, self.outputs = {}, {}
        self.reuse = reuse_buffers and os.environ.get("VIDC_NO_BUFFER_REUSE", "0") != "1"      # (debug knob of tools/dbg_lanes.py: every program tensor in storage of its own)
        self.stream_id = 0
        self.wait_mask = 0
        self.handle = None
        self.flops = 0           # executed conv FLOPs (2*M*N*K summed; Winograd layers count their transform-domain GEMMs)
        self.direct_flops = 0    # the executed layers in direct form (what self.flops was before the Winograd layers)
        self.ref_flops = 0       # the same layers in the reference's formulation (1x1 convs AFTER their upsample: SURVEY §8d)
        self._keep = []
        self._split_cache = {}   # (buf, ch_off, channels) -> buffer id of the split-bf16 image
        self.cuts = []           # op indices where a new segment starts (see cut()); filled by finalize()
        self._cut_markers = []
        self.mode = mode if mode is not None else precision_mode()
        self.winograd = winograd

    # ---- buffers ----------------------------------------------------------------------------------------
    def _new_buf(self, elems, pinned=False):
        self.buf_elems.append(int(elems))
        if pinned:
            self.pinned.add(len(self.buf_elems) - 1)
        return len(self.buf_elems) - 1

    def input_nchw(self, name, Cc, H, W):
        t = T(self._new_buf(self.B * Cc * H * W, True), self.B, H, W, Cc, nchw=True)
        self.inputs[name] = t
        return t

    def input_raw(self, name, elems):
        t = T(self._new_buf(elems, True), 1, 1, 1, elems)
        self.inputs[name] = t
        return t

    def mark_output(self, name, t):
        self.pinned.add(t.buf)
        self.outputs[name] = t

    def nhwc(self, H, W, Cc, G=1):
        return T(self._new_buf(self.B * H * W * G * Cc), self.B, H, W, Cc, G)

    def cut(self):
        """Starts a new segment: the ops recorded so far and the ops that follow can be captured / launched separately
        (capture_segments / launch_segment), with host-enqueued work in between."""
        assert self.ops, "cut() before the first op"
        self._cut_markers.append(self.ops[-1])      # identity of the last op of the segment (op indices shift in _fuse_splits)

    def on_stream(self, sid, wait_mask=0):
        self.stream_id, self.wait_mask = sid, wait_mask

    def _emit(self, kind, reads, writes, **kw):
        kw.update(stream_id=self.stream_id, wait_mask=self.wait_mask)
        self.wait_mask = 0
        self.ops.append((kind, [t.buf for t in reads if t is not None], [t.buf for t in writes], kw))

    # ---- ops ---------------------------------------------------------------------------------------------
    def conv(self, x, key, bn=None, relu=False, stride=1, padding=0, bn2=None, relu2=False, residual=None,
             relu_after_residual=False, out=None, accumulate=False, ref_flops_scale=1.0, dilation=1):
        keys = _keys(key)
        G = len(keys)
        assert x.G == G and not x.nchw
        w0 = self.ws.raw(keys[0] + ".weight")
        co, ci, kh, kw = w0.shape
        assert ci == x.C, "conv %s expects Cin=%d, got %d" % (keys[0], ci, x.C)
        Ho = (x.H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
        Wo = (x.W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
        y = out if out is not None else self.nhwc(Ho, Wo, co, G)
        assert (y.H, y.W, y.C, y.G) == (Ho, Wo, co, G)
        self._split_cache = {k: v for k, v in self._split_cache.items() if k[0] != y.buf}     # y is (re)written
        flags = (L.RELU1 if relu else 0) | (L.AFFINE2 if bn2 is not None else 0) | (L.RELU2 if relu2 else 0)
        if residual is not None:
            assert (residual.H, residual.W, residual.C, residual.G) == (Ho, Wo, co, G)
            flags |= L.RESIDUAL | (L.RELU3 if relu_after_residual else 0)
        if accumulate:
            flags |= L.ACCUM
        flops = 2 * self.B * Ho * Wo * co * ci * kh * kw * G
        wm = 0
        if residual is None and not accumulate:
            wm = winograd_choice(self.B, x.H, x.W, co, ci, kh, kw, stride, padding, dilation, G, mode=self.winograd, precision=self.mode)
        if wm:
            self.ref_flops += int(round(flops * ref_flops_scale))
            self.direct_flops += flops
            # F(4 x 4) in ONE launch (csrc/wfused.hip): where the measured table says 5, or -- VIDC_WINO_FUSED=<tiles>, A-B runs -- every fp32
            # F(4 x 4) layer of at most that many tiles and VIDC_WINO_FUSED_MAXC (256) input channels; VIDC_WINO_FUSED=0 keeps the three launches everywhere
            knob = os.environ.get("VIDC_WINO_FUSED")
            fusable = self.mode == "fp32" and ci % 16 == 0 and not (flags & ~(L.RELU1 | L.AFFINE2 | L.RELU2))
            forced = (wm in (4, 5) and knob is not None and int(knob) > 0 and self.B * (-(-Ho // 4)) * (-(-Wo // 4)) <= int(knob)
                      and ci <= int(os.environ.get("VIDC_WINO_FUSED_MAXC", "256")))
            if fusable and (forced or (wm == 5 and knob is None)):
                return self._conv_winograd_fused(x, y, keys, bn, bn2, flags, (co, ci, Ho, Wo))
            return self._conv_winograd(x, y, keys, bn, bn2, flags, 4 if wm == 5 else wm, (co, ci, Ho, Wo))
        self.flops += flops
        self.direct_flops += flops
        self.ref_flops += int(round(flops * ref_flops_scale))
        # precision: measured table entry if there is one, else by size
        sig = "M%d_N%d_K%d_k%ds%d_G%d" % (self.B * Ho * Wo, co, kh * kw * ci, kh, stride, G)
        prec = L.PREC_FP32
        if self.mode == "mixed":
            ent = tuning_table().get(sig)
            prec = ent[2] if (ent is not None and len(ent) > 2) else default_precision(flops)
        xin = self.split(x) if prec == L.PREC_BF16X3 else x
        self._emit("conv", [xin, residual, y if accumulate else None], [y], x=xin, y=y, keys=keys, precision=prec,
                   bn=_keys(bn) if bn is not None else None, bn2=_keys(bn2) if bn2 is not None else None,
                   residual=residual, flags=flags, stride=stride, pad=padding, geom=(co, ci, kh, kw, Ho, Wo), dilation=dilation)
        return y

    def _conv_winograd(self, x, y, keys, bn, bn2, flags, m, geom):
        """3x3 / stride 1 / pad 1 conv + BN + ReLU as Winograd F(m x m, 3x3): input transform -> ONE grouped 1x1 GEMM launch with
        (m+2)^2 * G groups on the MFMA kernel -> output transform with the conv's epilogue (csrc/winograd.hip)."""
        co, ci, Ho, Wo = geom
        G = len(keys)
        a2 = (m + 2) * (m + 2)
        tiles = self.B * (-(-Ho // m)) * (-(-Wo // m))
        gflops = 2 * tiles * a2 * G * co * ci
        self.flops += gflops
        sig = "M%d_N%d_K%d_k1s1_G%d" % (tiles, co, ci, a2 * G)
        prec = L.PREC_FP32
        if self.mode == "mixed":
            ent = tuning_table().get(sig)
            prec = ent[2] if (ent is not None and len(ent) > 2) else default_precision(gflops)
        V = T(self._new_buf(tiles * a2 * G * ci), 1, 1, tiles, ci, a2 * G)
        Mm = T(self._new_buf(tiles * a2 * G * co), 1, 1, tiles, co, a2 * G)
        self._emit("wino_in", [x], [V], x=x, v=V, m=m, cin=ci, split=int(prec == L.PREC_BF16X3))
        self._emit("conv", [V], [Mm], x=V, y=Mm, keys=keys, precision=prec, bn=None, bn2=None, residual=None, flags=0, stride=1, pad=0,
                   geom=(co, ci, 1, 1, 1, tiles), dilation=1, wino=m)
        self._emit("wino_out", [Mm], [y], mm=Mm, y=y, keys=keys, bn=_keys(bn) if bn is not None else None,
                   bn2=_keys(bn2) if bn2 is not None else None, flags=flags, m=m, cout=co)
        return y

    def _conv_winograd_fused(self, x, y, keys, bn, bn2, flags, geom):
        """The same layer as ONE launch (csrc/wfused.hip, tile VIDC_TILE_WINO4_FUSED): fp32, F(4 x 4), no V / M buffers.  Only behind VIDC_WINO_FUSED
        (measured equal to the three launches alone)."""
        co, ci, Ho, Wo = geom
        tiles = self.B * (-(-Ho // 4)) * (-(-Wo // 4))
        self.flops += 2 * tiles * 36 * len(keys) * co * ci
        self._emit("conv", [x], [y], x=x, y=y, keys=keys, precision=L.PREC_FP32, bn=_keys(bn) if bn is not None else None,
                   bn2=_keys(bn2) if bn2 is not None else None, residual=None, flags=flags, stride=1, pad=1, geom=(co, ci, 3, 3, Ho, Wo), dilation=1,
                   wino_fused=tiles)
        return y

    def split(self, x):
        """Split-bf16 image of an fp32 activation tensor (same geometry / bytes), made once per tensor."""
        ck = (x.buf, x.ch_off, x.C * x.G)
        if ck not in self._split_cache:
            assert not x.nchw and (x.C * x.G) % 32 == 0
            # a channel slice of a tensor whose split image exists is a slice of that image: the split layout keeps every
            # 32-channel unit in place ([32 x hi | 32 x lo] = the same 128 bytes), so only ch_off (a multiple of 32) moves
            for (b0, off0, ch0), (sbuf, ld0) in self._split_cache.items():
                if b0 == x.buf and off0 <= x.ch_off and x.ch_off + x.C * x.G <= off0 + ch0 and (x.ch_off - off0) % 32 == 0 and ld0 == ch0:
                    return T(sbuf, x.B, x.H, x.W, x.C, x.G, ld=ld0, ch_off=x.ch_off - off0)
            y = T(self._new_buf(self.B * x.H * x.W * x.C * x.G), x.B, x.H, x.W, x.C, x.G)
            self._emit("split", [x], [y], x=x, y=y)
            self._split_cache[ck] = (y.buf, x.C * x.G)
        # same storage, the caller's view of it (grouped and channel-concatenated views share one split image)
        sbuf, ld0 = self._split_cache[ck]
        return T(sbuf, x.B, x.H, x.W, x.C, x.G, ld=ld0)

    def linear(self, x, key, relu=False):
        """nn.Linear on `x.view(B, -1)` of the reference's NCHW tensor (surface_normal_dorn.py:23-24), as a 1x1 conv over the
        NHWC buffer viewed as one pixel with H*W*C channels; the weight columns are re-ordered (c,h,w) -> (h,w,c) once."""
        assert x.G == 1 and x.ld == x.C and x.ch_off == 0 and not x.nchw
        Cc, Hh, Ww = x.C, x.H, x.W
        vkey = "%s@hwc%dx%dx%d" % (key, Hh, Ww, Cc)

        def relayout(sd, key=key):
            w = sd[key + ".weight"]
            return w.view(w.shape[0], Cc, Hh, Ww).permute(0, 2, 3, 1).reshape(w.shape[0], Hh * Ww * Cc, 1, 1).contiguous()

        self.ws.add_virtual(vkey + ".weight", relayout)
        self.ws.add_virtual(vkey + ".bias", lambda sd, key=key: sd[key + ".bias"])
        flat = T(x.buf, x.B, 1, 1, Hh * Ww * Cc)
        return self.conv(flat, vkey, relu=relu)

    def avgpool(self, x, kernel, stride, padding):
        kh, kw = kernel
        sh, sw = stride
        ph, pw = padding
        Ho, Wo = (x.H + 2 * ph - kh) // sh + 1, (x.W + 2 * pw - kw) // sw + 1
        y = self.nhwc(Ho, Wo, x.C, x.G)
        self._emit("avgpool", [x], [y], x=x, y=y, geom=(kh, kw, sh, sw, ph, pw))
        return y

    def normalize_nchw(self, x):
        assert x.nchw
        y = T(self._new_buf(x.B * x.C * x.H * x.W), x.B, x.H, x.W, x.C, nchw=True)
        self._emit("normalize", [x], [y], x=x, y=y)
        return y

    def stem_conv(self, xs, key, relu=True, x_is_nchw=True):
        """xs: one NCHW tensor or a list (one per group, Cin may differ: 3,3,1)."""
        keys = _keys(key)
        xs = xs if isinstance(xs, (list, tuple)) else [xs]
        G = len(keys)
        assert len(xs) == G and x_is_nchw
        co = self.ws.raw(keys[0] + ".weight").shape[0]
        H, W = xs[0].H, xs[0].W
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = self.nhwc(Ho, Wo, co, G)
        for g in range(G):
            self._emit("stem", [xs[g]], [y], x=xs[g], y=y, key=keys[g], g=g, relu=relu)
        return y

    def det_im2col(self, image, Hp, Wp, mean_bgr):
        """The detector's input transform fused with the im2col of its 7x7/s2 stem (vidc_det_stem_im2col): image NCHW in [0,1] ->
        NHWC (B, Hp/2, Wp/2, 160)."""
        assert image.nchw and image.C == 3
        y = self.nhwc(Hp // 2, Wp // 2, 160)
        self._emit("det_im2col", [image], [y], x=image, y=y, geom=(image.H, image.W, Hp, Wp), mean=tuple(float(v) for v in mean_bgr))
        return y

    def nearest2x(self, x):
        y = self.nhwc(2 * x.H, 2 * x.W, x.C, x.G)
        self._emit("nearest2x", [x], [y], x=x, y=y)
        return y

    def mask_scale(self, x, image):
        """x * (r+g+b > 1e-2 of `image`, nearest-resized to x): the use_mask branch of surface_normal.py:150-162.  `image`: NCHW."""
        assert image.nchw and image.C == 3 and not x.nchw
        y = self.nhwc(x.H, x.W, x.C, x.G)
        self._emit("mask", [x, image], [y], x=x, y=y, image=image)
        return y

    def maxpool(self, x):
        Ho, Wo = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
        y = self.nhwc(Ho, Wo, x.C, x.G)
        self._emit("maxpool", [x], [y], x=x, y=y)
        return y

    def upsample(self, x, size, relu=False, into=None, out=None, sum_groups=False):
        """UpsamplingBilinear2d(size).  `into`: accumulate (+=) into an existing tensor; `out`: write (=) into an existing tensor
        (e.g. a channel slice of a concat buffer) instead of a new one.  sum_groups: x is a grouped tensor whose G upsampled (ReLU'd)
        groups are added up, group 0 first -- z2 + z3 + z4 of the decoders in one launch."""
        if sum_groups:
            assert into is not None and x.G > 1 and (into.H, into.W, into.C * into.G) == (size[0], size[1], x.C)
            self._split_cache = {k: v for k, v in self._split_cache.items() if k[0] != into.buf}
            self._emit("upsample", [x, into], [into], x=x, y=into, flags=(L.UP_RELU if relu else 0) | L.UP_ACCUM | (x.G << 8), sum_groups=x.G)
            return into
        y = into if into is not None else (out if out is not None else self.nhwc(size[0], size[1], x.C, x.G))
        assert (y.H, y.W, y.C * y.G) == (size[0], size[1], x.C * x.G)
        flags = (L.UP_RELU if relu else 0) | (L.UP_ACCUM if into is not None else 0)
        self._split_cache = {k: v for k, v in self._split_cache.items() if k[0] != y.buf}
        self._emit("upsample", [x, into], [y], x=x, y=y, flags=flags)
        return y

    def head(self, x, key, pad, out_size, relu):
        """1x1 conv to <=4 channels (+pad) -> upsample to out_size -> [relu]; returns the NCHW output tensor."""
        w = self.ws.raw(key + ".weight")
        co, ci = w.shape[0], w.shape[1]
        assert ci == x.C * x.G
        hp, wp = x.H + 2 * pad, x.W + 2 * pad
        low = T(self._new_buf(self.B * co * hp * wp), self.B, hp, wp, co, nchw=True)
        y = T(self._new_buf(self.B * co * out_size[0] * out_size[1]), self.B, out_size[0], out_size[1], co, nchw=True)
        self._emit("head", [x], [low, y], x=x, low=low, y=y, key=key, pad=pad, relu=relu)
        return y, low

    def warp_params(self, g, a, intr, kinv):
        p = T(self._new_buf(self.B * L.WARP_PARAMS), 1, 1, 1, self.B * L.WARP_PARAMS)
        self._emit("warp_params", [g, a, kinv], [p], g=g, a=a, kinv=kinv, p=p, intr=intr)
        return p

    def warp_fwd(self, x, params, intr, align_corners):
        y = T(self._new_buf(x.B * x.C * x.H * x.W), x.B, x.H, x.W, x.C, nchw=True)
        self._emit("warp_fwd", [x, params], [y], x=x, p=params, y=y, intr=intr, ac=int(align_corners))
        return y

    def warp_inv(self, x, params, intr, align_corners, normalize=True, out=None):
        """`out`: an existing NCHW (B,3,H,W) tensor to write into (e.g. the input buffer a later tick reads the normals from)."""
        y = out if out is not None else T(self._new_buf(x.B * 3 * x.H * x.W), x.B, x.H, x.W, 3, nchw=True)
        assert y.nchw and (y.B, y.C, y.H, y.W) == (x.B, 3, x.H, x.W)
        self._emit("warp_inv", [x, params], [y], x=x, p=params, y=y, intr=intr, ac=int(align_corners), norm=int(normalize))
        return y

    def copy(self, src, dst):
        self._emit("copy", [src], [dst], src=src, dst=dst)

    # ---- planning ----------------------------------------------------------------------------------------
    def _fuse_splits(self):
        """A split op whose input was just produced by a fused conv -- or by a stem / max-pool / upsample kernel -- is folded into
        that producer (VIDC_SPLIT_OUT / the y_split argument of the glue kernels); when nothing else reads the fp32 result
        the producer does not store it at all (VIDC_NO_F32_OUT, VIDC_UP_NO_F32_OUT, y = NULL)."""
        n = len(self.ops)
        drop = set()
        for i, (kind, _r, _w, kw) in enumerate(self.ops):
            if kind != "split":
                continue
            xs = kw["x"]
            j = next((t for t in range(i - 1, -1, -1) if xs.buf in self.ops[t][2]), None)
            if j is None or self.ops[j][0] not in ("conv", "maxpool", "upsample", "stem", "wino_out"):
                continue
            pkind, pk = self.ops[j][0], self.ops[j][3]
            y = pk["y"]
            if pkind == "stem":
                # G launches write the channel slices of one tensor: all of them get the image of the whole tensor
                writers = [t for t in range(j + 1) if self.ops[t][0] == "stem" and self.ops[t][3]["y"].buf == y.buf]
                if xs.ch_off != 0 or xs.C * xs.G != y.C * y.G or y.ld != y.C * y.G or y.ld % 32 or y.C % 32 or \
                        any(self.ops[t][3].get("split_out") is not None for t in writers) or len(writers) != y.G:
                    continue
                for t in writers:
                    self.ops[t][3]["split_out"] = kw["y"]
                    self.ops[t][2].append(kw["y"].buf)
                drop.add(i)
                continue
            # general case: the tensor is written by one producer, or slice by slice by several (a concat buffer filled by a conv and
            # an upsample, say); every producer writes its slice of the split image.  The slices must tile the split's channel range.
            last = {}                                        # channel slice -> its LAST writer (earlier ones are superseded)
            foreign = False
            for t in range(i):
                if xs.buf in self.ops[t][2]:
                    if self.ops[t][0] not in ("conv", "maxpool", "upsample", "wino_out"):
                        foreign = True
                        continue
                    yt = self.ops[t][3]["y"]
                    if yt.buf == xs.buf:
                        last[(yt.ch_off, yt.C * yt.G)] = t
            writers = sorted(last.values())
            lo, hi = xs.ch_off, xs.ch_off + xs.C * xs.G
            ok = bool(writers) and not foreign and xs.ld % 32 == 0 and lo % 32 == 0
            cover = []
            for t in writers:
                yt = self.ops[t][3]["y"]
                ok = ok and yt.ld == xs.ld and yt.ch_off % 32 == 0 and (yt.C * yt.G) % 32 == 0 and lo <= yt.ch_off and \
                    yt.ch_off + yt.C * yt.G <= hi and self.ops[t][3].get("split_out") is None
                cover.append((yt.ch_off, yt.ch_off + yt.C * yt.G))
            cover.sort()
            ok = ok and cover[0][0] == lo and cover[-1][1] == hi and all(cover[k][1] == cover[k + 1][0] for k in range(len(cover) - 1))
            if not ok:
                continue
            simg = kw["y"]                                   # the split image: compact, ld = xs.C * xs.G
            for t in writers:
                yt = self.ops[t][3]["y"]
                self.ops[t][3]["split_out"] = T(simg.buf, yt.B, yt.H, yt.W, yt.C, yt.G, ld=simg.C * simg.G, ch_off=yt.ch_off - lo)
                if self.ops[t][0] in ("conv", "wino_out"):
                    self.ops[t][3]["flags"] |= L.SPLIT_OUT
                self.ops[t][2].append(simg.buf)
            drop.add(i)
        for j, (kind, _r, _w, kw) in enumerate(self.ops):
            if kind in ("conv", "maxpool", "upsample", "stem", "wino_out") and kw.get("split_out") is not None and kw["y"].buf not in self.pinned:
                yb = kw["y"].buf
                used = any(t not in drop and yb in self.ops[t][1] for t in range(j + 1, n))
                # a later writer of the same buffer that accumulates into it needs the fp32 values too (reads cover that); a
                # later plain writer of ANOTHER slice of the buffer (stem groups) does not
                if not used:
                    if kind in ("conv", "wino_out"):
                        kw["flags"] |= L.NO_F32_OUT
                    elif kind == "upsample":
                        kw["flags"] |= L.UP_NO_F32_OUT
                    else:
                        kw["no_f32"] = True
        self.n_fused_splits = len(drop)
        self.ops = [op for i, op in enumerate(self.ops) if i not in drop]

    def _fuse_warp_into_stem(self):
        """A stem conv whose input is the output of a forward warp that nothing else reads gathers its input through the warp itself
        (vidc_stem_conv3x3s2_warped: the tap sets of warp_fwd_kernel, bit for bit) and the warp launch + the warped image go away
        (VERDICT r4 item 9).  Not when the warped image has another reader (the use_mask branch, surface_normal.py:150-162) or is an output."""
        drop = set()
        for i, (kind, reads, _w, kw) in enumerate(self.ops):
            if kind != "stem" or kw["x"].C != 3:
                continue
            xb = kw["x"].buf
            j = next((t for t in range(i - 1, -1, -1) if xb in self.ops[t][2]), None)
            if j is None or self.ops[j][0] != "warp_fwd" or xb in self.pinned:
                continue
            if any(t != i and t not in drop and xb in self.ops[t][1] for t in range(len(self.ops))):
                continue
            if any(id(self.ops[t]) in {id(mk) for mk in self._cut_markers} for t in range(j, i)):      # (a segment boundary in between)
                continue
            wkw = self.ops[j][3]
            kw["warp"] = {"x": wkw["x"], "p": wkw["p"], "intr": wkw["intr"], "ac": wkw["ac"]}
            self.ops[i] = (kind, [b for b in reads if b != xb] + [wkw["x"].buf, wkw["p"].buf], _w, kw)
            drop.add(j)
        if drop:
            self._cut_markers = [mk for mk in self._cut_markers]      # (markers are op identities: the dropped warp ops are never markers, checked above)
            self.ops = [op for t, op in enumerate(self.ops) if t not in drop]
        self.n_fused_warps = len(drop)

    def _fill_conv_desc(self, d, kw, addr, dry_run):
        """Fills one vidc_conv_desc from a recorded conv; returns the op's display name."""
        lib = L.lib()
        x, y, keys = kw["x"], kw["y"], kw["keys"]
        co, ci, kh, kwid, Ho, Wo = kw["geom"]
        prec = kw["precision"]
        wm = kw.get("wino", 0)
        wf = kw.get("wino_fused", 0)
        if wm:
            wp = self.ws.packed_winograd([k for k in keys], wm, dry_run, prec)
            s1, b1 = self.ws.identity_affine(co, wp.device)
        elif wf:
            wp = self.ws.packed_winograd_fused([k for k in keys], dry_run)
            s1, b1 = self.ws.affine(list(keys), list(kw["bn"]) if kw["bn"] is not None else None)
        else:
            wp = self.ws.packed([k for k in keys], dry_run, prec)
            s1, b1 = self.ws.affine(list(keys), list(kw["bn"]) if kw["bn"] is not None else None)
        d.x, d.w, d.y = addr(x), wp.data_ptr(), addr(y)
        d.scale1, d.shift1 = s1.data_ptr(), b1.data_ptr()
        if kw["bn2"] is not None:
            s2, b2 = self.ws.affine([None] * len(keys), list(kw["bn2"]))
            d.scale2, d.shift2 = s2.data_ptr(), b2.data_ptr()
            self._keep += [s2, b2]
        r = kw["residual"]
        if r is not None:
            d.residual, d.ldr, d.r_gs = addr(r), r.ld, r.C
        d.B, d.H, d.W, d.Cin, d.ldx = x.B, x.H, x.W, ci, x.ld
        d.Ho, d.Wo, d.Cout, d.ldy = Ho, Wo, co, y.ld
        d.KH, d.KW, d.stride, d.pad = kh, kwid, kw["stride"], kw["pad"]
        d.dilation = kw.get("dilation", 1)
        d.flags, d.groups = kw["flags"], len(keys)
        if kw.get("split_out") is not None:
            d.y_split = addr(kw["split_out"])
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = x.C, co * kh * kwid * ci, y.C, co
        if wm:          # (m+2)^2 transform-domain GEMMs per group of the layer, identity epilogue shared by all of them
            d.groups, d.p_gs = len(keys) * (wm + 2) * (wm + 2), 0
        d.tile, d.splitk, d.precision = 0, 1, prec
        sig = conv_signature(d)
        if wf:          # one launch: the transformed weights of a group are 36 Cout Cin floats; shown with the signature of the products it executes
            d.w_gs, d.tile = 36 * co * ci, L.TILE_WINO4_FUSED
            self._keep += [wp, s1, b1]
            return "conv:%s@wino4f:%s:sk1:fp32 M%d_N%d_K%d_k1s1_G%d flags=0x%x" % (keys[0], L.TILE_NAMES[d.tile], wf, co, ci, 36 * len(keys), d.flags)
        if os.environ.get("VIDC_FORCE_TILE"):           # (tests / A-B runs: one tiling for every conv)
            d.tile, d.splitk = int(os.environ["VIDC_FORCE_TILE"]), 1
        else:
            ent = tuning_table().get(sig)
            if ent is not None and (len(ent) < 5 or prec == ent[2]):
                d.tile, d.splitk = ent[0], ent[1]
            elif ent is not None and len(ent) >= 5:          # table holds the best fp32 config as well: [t, sk, prec, t32, sk32]
                d.tile, d.splitk = ent[3], ent[4]
            else:
                L.check(lib.vidc_conv2d_plan(C.byref(d)), "conv plan")
        if os.environ.get("VIDC_TILE_REMAP"):          # experiment knob: "6:28,10:28" replaces tile 6 and 10 by 28 wherever the table picked them
            remap = dict((int(a), int(b)) for a, b in (kv.split(":") for kv in os.environ["VIDC_TILE_REMAP"].split(",") if kv))
            d.tile = remap.get(d.tile, d.tile)
        if os.environ.get("VIDC_LDS_CAP_KB"):          # experiment knob (tools/dual_stream_bench.py): tilings that leave room for a second
            d.tile = _capped_tile(d.tile, int(os.environ["VIDC_LDS_CAP_KB"]))      # workgroup of another stream on the CU
        self._keep += [wp, s1, b1]
        return "conv:%s%s:%s:sk%d:%s %s flags=0x%x" % (keys[0], "@wino%d" % wm if wm else "", L.TILE_NAMES[d.tile], d.splitk, "bf16x3" if prec else "fp32", sig, d.flags)

    def _plan_buffers(self):
        n = len(self.buf_elems)
        multi_stream = any(kw["stream_id"] != 0 for _, _, _, kw in self.ops)
        storage = [None] * n
        if not self.reuse or multi_stream:
            for b in range(n):
                storage[b] = torch.zeros(max(self.buf_elems[b], 4), dtype=torch.float32, device=self.device)
            return storage
        last = [-1] * n
        first = [None] * n
        for i, (_, reads, writes, _) in enumerate(self.ops):
            for b in reads + writes:
                last[b] = i
                if first[b] is None:
                    first[b] = i
        free = []   # (elems, tensor)
        expire = {}
        for b in range(n):
            if b in self.pinned or first[b] is None:
                storage[b] = torch.zeros(max(self.buf_elems[b], 4), dtype=torch.float32, device=self.device)
            else:
                expire.setdefault(last[b], []).append(b)
        for i, (_, reads, writes, _) in enumerate(self.ops):
            for b in writes + reads:
                if storage[b] is None:
                    need = self.buf_elems[b]
                    best = None
                    for j, (el, _t) in enumerate(free):
                        if el >= need and (best is None or el < free[best][0]):
                            best = j
                    if best is not None:
                        storage[b] = free.pop(best)[1]
                    else:
                        storage[b] = torch.zeros(max(need, 4), dtype=torch.float32, device=self.device)
            for b in expire.get(i, []):
                free.append((storage[b].numel(), storage[b]))
        return storage

    def finalize(self, dry_run=False):
        """Plans buffers and builds the C op array.  dry_run=True stops before any HIP call (host-logic tests on CPU)."""
        lib = L.lib()
        # Opt-in (VIDC_FUSE_WARP=1): bit-identical, one launch and 2 x 3HW x 4 bytes less per frame, no measurable gain (< 0.2 % of the tick).  (The wrong
        # frames round 5 saw with it were an execution defect of a packed-fp32 instruction beside bf16 MFMAs, not this kernel's loads: DESIGN 4.5.)
        if os.environ.get("VIDC_FUSE_WARP", "0") == "1":
            self._fuse_warp_into_stem()
        else:
            self.n_fused_warps = 0
        if os.environ.get("VIDC_FUSE_SPLIT", "1") == "1":
            self._fuse_splits()
        self.cuts = [next(i for i, op in enumerate(self.ops) if op is mk) + 1 for mk in self._cut_markers]
        assert len(self.cuts) < L.MAX_SEGMENTS and self.cuts == sorted(set(self.cuts))
        storage = self._plan_buffers()
        self.storage = storage
        self.bytes_allocated = sum({t.data_ptr(): t.numel() * 4 for t in storage if t is not None}.values())

        def addr(t, extra=0):
            return storage[t.buf].data_ptr() + 4 * (t.ch_off + extra)

        ops = (L.Op * sum(1 for _ in self.ops))()
        ws_need = {}
        conv_ops = []
        self.op_names = []
        for i, (kind, _r, _w, kw) in enumerate(self.ops):
            op = ops[i]
            op.stream_id, op.wait_mask = kw["stream_id"], kw["wait_mask"]
            g = op.u.g
            if kind == "conv":
                op.kind = L.OP_CONV
                name = self._fill_conv_desc(op.u.conv, kw, addr, dry_run)
                need = lib.vidc_conv2d_workspace_bytes(C.byref(op.u.conv))
                if need:
                    ws_need[op.stream_id] = max(ws_need.get(op.stream_id, 0), need)
                conv_ops.append(op)
                self.op_names.append(name)
            elif kind == "stem":
                x, y = kw["x"], kw["y"]
                w = self.ws.raw(kw["key"] + ".weight").contiguous()
                self._keep.append(w)
                op.kind = L.OP_STEM
                g.p[0], g.p[1], g.p[2] = addr(x), w.data_ptr(), (0 if kw.get("no_f32") else addr(y, kw["g"] * y.C))
                for j, v in enumerate((x.B, x.C, x.H, x.W, y.C, y.ld, int(kw["relu"]), kw["g"] * y.C)):
                    g.i[j] = v
                if kw.get("split_out") is not None:
                    g.p[3] = addr(kw["split_out"])
                wf = kw.get("warp")
                if wf is not None:             # the input is gathered through the forward warp: x = the unwarped image, p[4] = the parameter records
                    g.p[0], g.p[4] = addr(wf["x"]), addr(wf["p"])
                    g.f[0], g.f[1], g.i[8] = wf["intr"].cx, wf["intr"].cy, wf["ac"]
                self.op_names.append("stem:" + kw["key"] + ("+warp" if wf is not None else ""))
            elif kind == "maxpool":
                x, y = kw["x"], kw["y"]
                op.kind = L.OP_MAXPOOL
                g.p[0], g.p[1] = addr(x), (0 if kw.get("no_f32") else addr(y))
                for j, v in enumerate((x.B, x.H, x.W, x.C * x.G, x.ld, y.ld)):
                    g.i[j] = v
                if kw.get("split_out") is not None:
                    g.p[2] = addr(kw["split_out"])
                self.op_names.append("maxpool")
            elif kind == "upsample":
                x, y = kw["x"], kw["y"]
                op.kind = L.OP_UPSAMPLE
                g.p[0], g.p[1] = addr(x), addr(y)
                for j, v in enumerate((x.B, x.H, x.W, (x.C if kw.get("sum_groups") else x.C * x.G), x.ld, y.H, y.W, y.ld, kw["flags"])):
                    g.i[j] = v
                if kw.get("split_out") is not None:
                    g.p[2] = addr(kw["split_out"])
                self.op_names.append("upsample:%dx%dx%d->%dx%d" % (x.H, x.W, x.C * x.G, y.H, y.W))
            elif kind == "head":
                x, low, y = kw["x"], kw["low"], kw["y"]
                w = self.ws.raw(kw["key"] + ".weight").reshape(y.C, -1).contiguous()
                b = self.ws.raw(kw["key"] + ".bias").contiguous()
                self._keep += [w, b]
                op.kind = L.OP_HEAD
                g.p[0], g.p[1], g.p[2], g.p[3], g.p[4] = addr(x), w.data_ptr(), b.data_ptr(), addr(low), addr(y)
                for j, v in enumerate((x.B, x.H, x.W, x.C * x.G, x.ld, y.C, kw["pad"], y.H, y.W, int(kw["relu"]))):
                    g.i[j] = v
                self.op_names.append("head:" + kw["key"])
            elif kind == "warp_params":
                it = kw["intr"]
                op.kind = L.OP_WARP_PARAMS
                g.p[0], g.p[1], g.p[2], g.p[3] = addr(kw["g"]), addr(kw["a"]), addr(kw["kinv"]), addr(kw["p"])
                g.i[0], g.i[1], g.i[2] = self.B, it.W, it.H
                g.f[0], g.f[1], g.f[2], g.f[3] = it.fx, it.fy, it.cx, it.cy
                self.op_names.append("warp_params")
            elif kind == "warp_fwd":
                x, it = kw["x"], kw["intr"]
                op.kind = L.OP_WARP_FWD
                g.p[0], g.p[1], g.p[2] = addr(x), addr(kw["p"]), addr(kw["y"])
                for j, v in enumerate((x.B, x.C, x.H, x.W, kw["ac"])):
                    g.i[j] = v
                g.f[0], g.f[1] = it.cx, it.cy
                self.op_names.append("warp_fwd")
            elif kind == "warp_inv":
                x, it = kw["x"], kw["intr"]
                op.kind = L.OP_WARP_INV
                g.p[0], g.p[1], g.p[2] = addr(x), addr(kw["p"]), addr(kw["y"])
                for j, v in enumerate((x.B, x.H, x.W, kw["ac"], kw["norm"])):
                    g.i[j] = v
                g.f[0], g.f[1] = it.cx, it.cy
                self.op_names.append("warp_inv_rot_norm")
            elif kind == "split":
                x, y = kw["x"], kw["y"]
                rows = x.B * x.H * x.W
                op.kind = L.OP_SPLIT
                g.p[0], g.p[1] = addr(x), addr(y)
                g.i[0], g.i[1], g.i[2], g.i[3] = rows & 0xFFFFFFFF, rows >> 32, x.C * x.G, x.ld
                self.op_names.append("split:%dx%d" % (rows, x.C * x.G))
            elif kind == "avgpool":
                x, y = kw["x"], kw["y"]
                op.kind = L.OP_AVGPOOL
                g.p[0], g.p[1] = addr(x), addr(y)
                for j, v in enumerate((x.B, x.H, x.W, x.C * x.G, x.ld) + tuple(kw["geom"]) + (y.ld,)):
                    g.i[j] = v
                self.op_names.append("avgpool")
            elif kind == "normalize":
                x, y = kw["x"], kw["y"]
                op.kind = L.OP_NORMALIZE
                g.p[0], g.p[1] = addr(x), addr(y)
                g.i[0], g.i[1], g.i[2] = x.B, x.C, x.H * x.W
                self.op_names.append("normalize")
            elif kind == "det_im2col":
                x, y = kw["x"], kw["y"]
                op.kind = L.OP_DET_IM2COL
                g.p[0], g.p[1] = addr(x), addr(y)
                for j, v in enumerate((x.B,) + tuple(kw["geom"])):
                    g.i[j] = v
                g.f[0], g.f[1], g.f[2] = kw["mean"]
                self.op_names.append("det_im2col")
            elif kind == "mask":
                x, y, im = kw["x"], kw["y"], kw["image"]
                op.kind = L.OP_MASK
                g.p[0], g.p[1], g.p[2] = addr(x), addr(im), addr(y)
                for j, v in enumerate((x.B, x.H, x.W, x.C * x.G, x.ld, y.ld, im.H, im.W)):
                    g.i[j] = v
                self.op_names.append("mask_scale")
            elif kind == "wino_in":
                x, v = kw["x"], kw["v"]
                op.kind = L.OP_WINO_IN
                g.p[0], g.p[1] = addr(x), addr(v)
                for j, val in enumerate((x.B, x.H, x.W, x.C * x.G, x.ld, kw["cin"], kw["m"], kw["split"], v.ld)):
                    g.i[j] = val
                self.op_names.append("wino_in:F%d:%dx%dx%d" % (kw["m"], x.H, x.W, x.C * x.G))
            elif kind == "wino_out":
                mm, y = kw["mm"], kw["y"]
                keys = kw["keys"]
                s1, b1 = self.ws.affine(list(keys), list(kw["bn"]) if kw["bn"] is not None else None)
                self._keep += [s1, b1]
                op.kind = L.OP_WINO_OUT
                g.p[0], g.p[1], g.p[3], g.p[4] = addr(mm), (0 if kw["flags"] & L.NO_F32_OUT else addr(y)), s1.data_ptr(), b1.data_ptr()
                if kw.get("split_out") is not None:
                    g.p[2] = addr(kw["split_out"])
                if kw["bn2"] is not None:
                    s2, b2 = self.ws.affine([None] * len(keys), list(kw["bn2"]))
                    g.p[5], g.p[6] = s2.data_ptr(), b2.data_ptr()
                    self._keep += [s2, b2]
                for j, val in enumerate((y.B, y.H, y.W, y.C * y.G, kw["cout"], y.ld, kw["m"], kw["flags"], mm.ld)):
                    g.i[j] = val
                self.op_names.append("wino_out:F%d:%dx%dx%d" % (kw["m"], y.H, y.W, y.C * y.G))
            elif kind == "nearest2x":
                x, y = kw["x"], kw["y"]
                op.kind = L.OP_NEAREST2X
                g.p[0], g.p[1] = addr(x), addr(y)
                for j, v in enumerate((x.B, x.H, x.W, x.C * x.G, x.ld, y.ld)):
                    g.i[j] = v
                self.op_names.append("nearest2x")
            elif kind == "copy":
                src, dst = kw["src"], kw["dst"]
                nbytes = min(self.buf_elems[src.buf], self.buf_elems[dst.buf]) * 4
                op.kind = L.OP_COPY
                g.p[0], g.p[1] = addr(src), addr(dst)
                g.i[0], g.i[1] = nbytes & 0xFFFFFFFF, nbytes >> 32
                self.op_names.append("copy")
            else:
                raise ValueError(kind)
        # split-K workspaces (one per stream id so concurrent convs never share partials); zeroed once: their heads hold the
        # per-tile ticket counters of the fused split-K reduction, which every launch leaves at zero (include/vidc.h)
        self.workspaces = {sid: torch.zeros(nb // 4 + 4, dtype=torch.float32, device=self.device) for sid, nb in ws_need.items()}
        for op in conv_ops:
            if op.u.conv.splitk > 1:
                op.u.conv.workspace = self.workspaces[op.stream_id].data_ptr()
        self.c_ops = ops
        self.captured = False
        if dry_run:
            return self
        h = C.c_void_p()
        L.check(lib.vidc_program_create(ops, len(ops), C.byref(h)), "program_create")
        self.handle = h
        self.captured = False
        # The pack / fold kernels above ran on the stream that was current while this program was built, and their results are
        # cached in the WeightStore: a program built later for another slot launches on ANOTHER stream and must not race them.
        # One host wait here (not on the hot path) orders every later consumer after the packing.
        torch.cuda.current_stream(self.device).synchronize()
        return self

    # ---- execution ----------------------------------------------------------------------------------------
    def tensor(self, t):
        """torch view of a program tensor (NCHW tensors come back shaped (B,C,H,W))."""
        s = self.storage[t.buf]
        if t.nchw:
            return s[: t.B * t.C * t.H * t.W].view(t.B, t.C, t.H, t.W)
        return s[: t.B * t.H * t.W * t.ld].view(t.B, t.H, t.W, t.ld)[..., t.ch_off: t.ch_off + t.C * t.G]

    def run(self, stream=None):
        L.check(L.lib().vidc_program_run(self.handle, stream if stream is not None else L.current_stream()), "program_run")

    def capture(self, stream=None):
        L.check(L.lib().vidc_program_capture(self.handle, stream if stream is not None else L.current_stream()), "program_capture")
        self.captured = True

    def segments(self):
        """[(begin, end)] op ranges separated by cut()."""
        edges = [0] + list(self.cuts) + [len(self.ops)]
        return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]

    def run_segment(self, k, stream=None):
        b, e = self.segments()[k]
        L.check(L.lib().vidc_program_run_range(self.handle, stream if stream is not None else L.current_stream(), b, e), "program_run_range")

    def capture_segments(self, stream=None):
        st = stream if stream is not None else L.current_stream()
        for k, (b, e) in enumerate(self.segments()):
            L.check(L.lib().vidc_program_capture_range(self.handle, st, b, e, k), "program_capture_range")
        self.captured = True

    def launch_segment(self, k, stream=None):
        L.check(L.lib().vidc_program_launch_segment(self.handle, stream if stream is not None else L.current_stream(), k), "program_launch_segment")

    def launch(self, stream=None):
        L.check(L.lib().vidc_program_launch(self.handle, stream if stream is not None else L.current_stream()), "program_launch")

    # ---- group-restricted variants of a segment ---------------------------------------------------------------------------
    def group_variant(self, name, segment, g_lo, g_hi, groups, keep_ungrouped):
        """A second native program over the SAME buffers, packed weights and tile choices: the ops of `segment` with every
        `groups`-group launch (fused convs, the max-pool over the grouped stem output, the per-group stem kernels) restricted to groups
        [g_lo, g_hi), and the segment's other ops kept or dropped.  pipeline.build_frame_program's tick runs the surface-normal pyramid
        (group 0) and the three depth-completion pyramids (groups 1..3) as 4-group launches; the first tick of a frame stream has no
        previous frame to complete and the drain tick no new one to start, so they run the variants (0, 1, keep) / (1, 4, drop) and the
        chip does the work of exactly n frames for n frames.  Per output element nothing changes: a group only selects base pointers
        (csrc/conv_mfma.hip: `a.x + g * a.x_gs` ...), while tile, split-K partition and K order -- which fix the fp32 summation order --
        are copied from the grouped launch.  So a frame's result is bit-identical whichever variant computed its pyramids
        (tests/test_hip_parity.py::test_run_interleaved_lanes_are_bit_identical)."""
        assert self.handle is not None and 0 <= g_lo < g_hi <= groups
        b, e = self.segments()[segment]
        picked = []
        for i in range(b, e):
            kind, _r, _w, kw = self.ops[i]
            if kind == "stem":
                if g_lo <= kw["g"] < g_hi:
                    picked.append(i)
                continue
            grouped = (kind in ("conv", "wino_out") and len(kw["keys"]) == groups) or (kind in ("maxpool", "wino_in") and kw["x"].G == groups)
            # (a split launch of its own -- VIDC_FUSE_SPLIT=0 -- over a grouped tensor has no row stride for its image: it runs whole in every variant)
            if grouped or keep_ungrouped or (kind == "split" and kw["x"].G == groups):
                picked.append(i)
        ops = (L.Op * len(picked))()
        for j, i in enumerate(picked):
            C.memmove(C.byref(ops[j]), C.byref(self.c_ops[i]), C.sizeof(L.Op))
            kind, _r, _w, kw = self.ops[i]
            if kind == "conv" and len(kw["keys"]) == groups:
                d = ops[j].u.conv
                a2 = (kw["wino"] + 2) ** 2 if kw.get("wino") else 1        # Winograd GEMMs: (m+2)^2 launch groups per group of the layer
                for field, gs in (("x", d.x_gs), ("w", d.w_gs), ("y", d.y_gs), ("scale1", d.p_gs), ("shift1", d.p_gs), ("scale2", d.p_gs),
                                  ("shift2", d.p_gs), ("residual", d.r_gs), ("y_split", d.y_gs)):     # every operand is 4 bytes per element
                    v = getattr(d, field)                                                              # (a split-bf16 unit = 32 x (hi, lo))
                    if v:
                        setattr(d, field, v + 4 * g_lo * gs * a2)
                d.groups = (g_hi - g_lo) * a2
            elif kind == "wino_in" and kw["x"].G == groups:        # x: group at channel g * cin; V rows: [gg][pos][cin] (row stride explicit)
                g = ops[j].u.g
                cin, a2 = kw["cin"], (kw["m"] + 2) ** 2
                g.p[0], g.p[1] = g.p[0] + 4 * g_lo * cin, g.p[1] + 4 * g_lo * a2 * cin
                g.i[3] = cin * (g_hi - g_lo)
            elif kind == "wino_out" and len(kw["keys"]) == groups:
                g = ops[j].u.g
                co, a2 = kw["cout"], (kw["m"] + 2) ** 2
                g.p[0] = g.p[0] + 4 * g_lo * a2 * co
                for k in (1, 2, 3, 4, 5, 6):                       # y, split image, scale / shift arrays: group g at element g * cout
                    if g.p[k]:
                        g.p[k] = g.p[k] + 4 * g_lo * co
                g.i[3] = co * (g_hi - g_lo)
            elif kind == "maxpool" and kw["x"].G == groups:
                g = ops[j].u.g
                cg = kw["x"].C
                for k in range(3):                            # x, y, split image: group g sits at channel offset g * C of every pixel row
                    if g.p[k]:
                        g.p[k] = g.p[k] + 4 * g_lo * cg
                g.i[3] = cg * (g_hi - g_lo)
        h = C.c_void_p()
        L.check(L.lib().vidc_program_create(ops, len(ops), C.byref(h)), "program_create (variant)")
        self._variants = getattr(self, "_variants", {})
        self._variants[name] = {"handle": h, "ops": ops, "captured": False, "indices": picked}
        return self._variants[name]

    def has_variant(self, name):
        return name in getattr(self, "_variants", {})

    def run_variant(self, name, stream=None):
        L.check(L.lib().vidc_program_run(self._variants[name]["handle"], stream if stream is not None else L.current_stream()), "variant_run")

    def capture_variant(self, name, stream=None):
        v = self._variants[name]
        L.check(L.lib().vidc_program_capture(v["handle"], stream if stream is not None else L.current_stream()), "variant_capture")
        v["captured"] = True

    def launch_variant(self, name, stream=None):
        v = self._variants[name]
        st = stream if stream is not None else L.current_stream()
        if v["captured"]:
            L.check(L.lib().vidc_program_launch(v["handle"], st), "variant_launch")
        else:
            L.check(L.lib().vidc_program_run(v["handle"], st), "variant_run")

    def time(self, iters=20, use_graph=False, per_op=False, stream=None):
        ms = (C.c_float * 1)()
        per = (C.c_float * len(self.ops))() if per_op else None
        L.check(L.lib().vidc_program_time(self.handle, stream if stream is not None else L.current_stream(), iters,
                                          int(use_graph), ms, per), "program_time")
        return (ms[0], list(per)) if per_op else ms[0]

    def __del__(self):
        try:
            if self.handle is not None:
                L.lib().vidc_program_destroy(self.handle)
            for v in getattr(self, "_variants", {}).values():
                L.lib().vidc_program_destroy(v["handle"])
        except Exception:
            pass
