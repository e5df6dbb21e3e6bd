"""Which neighbour makes the round-5 form of the fused stem (plain image loads, VIDC_DBG_STEM_LOADS=3 / 2) read wrong quads?
Victim lanes: [D2D copies of a fresh image + gravity] -> warp_params -> vidc_stem_conv3x3s2_warped, output compared on the device with the
quiescent result.  Noise lane: ONE kind of kernel in a loop on another stream.

    VIDC_DBG_STEM_LOADS=3 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 400
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from vi_depth_completion_amd import _lib as L
from vi_depth_completion_amd import ops
from vi_depth_completion_amd import synthetic as S

torch.set_grad_enabled(False)
DEV = "cuda"
B, H, W = 4, 240, 320


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", default="none")
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--victims", type=int, default=2)
    ap.add_argument("--per", type=int, default=2, help="noise launches per victim iteration")
    ap.add_argument("--victim", default="fused", help="fused | warp (the default path's warp_fwd + stem)")
    a = ap.parse_args()
    lib = L.lib()
    NI = 8
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    imgs = [torch.rand((B, 3, H, W), device=DEV, generator=gen) for _ in range(NI)]
    gr = [torch.nn.functional.normalize(torch.tensor([[0.05 * (i - 3) + 0.01 * b, 1.0, 0.1 * (i - 4)] for b in range(B)]), dim=1).to(DEV) for i in range(NI)]
    al = torch.tensor([[0.0, 1.0, 0.0]] * B, device=DEV)
    kinv = torch.tensor(np.linalg.inv(np.array([[202.0, 0, 159.94], [0, 202.0, 119.94], [0, 0, 1.0]])).astype(np.float32).reshape(-1), device=DEV)
    wt = S.normal01(5, "stem.w", (64, 3, 3, 3), scale=0.2).float().to(DEV)

    class Victim:
        def __init__(self):
            self.st = torch.cuda.Stream()
            self.x = torch.zeros(B, 3, H, W, device=DEV); self.x2 = torch.zeros_like(self.x); self.g = torch.zeros(B, 3, device=DEV)
            self.p = torch.zeros(B * 32, device=DEV); self.y = torch.zeros(B, H // 2, W // 2, 64, device=DEV); self.xw = torch.zeros_like(self.x)
            self.bad = []

        def ops_(self):
            s = self.st.cuda_stream
            L.check(lib.vidc_warp2dof_params(self.g.data_ptr(), al.data_ptr(), B, 202.0, 202.0, 159.94, 119.94, kinv.data_ptr(), W, H, self.p.data_ptr(), s), "p")
            if a.victim == "fused":
                L.check(lib.vidc_stem_conv3x3s2_warped(self.x.data_ptr(), self.p.data_ptr(), wt.data_ptr(), self.y.data_ptr(), B, H, W, 64, 64, 1, None, 0, 159.94, 119.94, 0, s), "stem")
            else:
                L.check(lib.vidc_warp2dof_fwd(self.x.data_ptr(), self.p.data_ptr(), self.xw.data_ptr(), B, 3, H, W, 159.94, 119.94, 0, s), "warp")
                L.check(lib.vidc_stem_conv3x3s2(self.xw.data_ptr(), wt.data_ptr(), self.y.data_ptr(), B, 3, H, W, 64, 64, 1, None, 0, s), "stem")

        def go(self, i, ref=None):
            with torch.cuda.stream(self.st):
                self.x2.copy_(self.x, non_blocking=True)                 # (the lane's dc_image.copy_(sn_image))
                for b in range(B):
                    self.x[b:b + 1].copy_(imgs[i][b:b + 1], non_blocking=True)
                self.g.copy_(gr[i], non_blocking=True)
                self.ops_()
                if ref is not None:
                    self.bad.append((self.y.view(torch.int32) != ref[i].view(torch.int32)).sum())
                    return None
                return self.y.clone()

    vs = [Victim() for _ in range(a.victims)]
    torch.cuda.synchronize()
    ref = []
    for i in range(NI):
        ref.append(vs[0].go(i)); torch.cuda.synchronize()

    # ---- noise ----------------------------------------------------------------------------------------------------------------------
    nst = torch.cuda.Stream()
    xa = torch.randn(4, 60, 80, 256, device=DEV)
    xb = torch.randn(4, 120, 160, 256, device=DEV)
    w33 = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
    one, zero = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    wp32 = ops.pack_conv_weight(w33)
    wpb = ops.pack_conv_weight_bf16x3(w33)
    xa_split = ops.split_bf16x3(xa)
    sp_a, sp_b = torch.empty_like(xa), torch.empty_like(xb)
    ximg = torch.rand(4, 3, H, W, device=DEV)
    sp_stem = torch.empty(4, 120, 160, 64, device=DEV)
    big0, big1 = torch.empty(16 << 20, device=DEV), torch.empty(16 << 20, device=DEV)
    torch.cuda.synchronize()

    def conv(prec, split_out=None):
        # (called with the split image prepared once, so the noise is the conv kernel alone)
        d = L.ConvDesc()
        x = xa_split if prec else xa
        wpk = wpb if prec else wp32
        y = conv.y
        d.x, d.w, d.y = L.ptr(x), L.ptr(wpk), L.ptr(y)
        d.scale1, d.shift1 = L.ptr(one), L.ptr(zero)
        d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 4, 60, 80, 256, 256, 60, 80, 256, 256
        d.KH, d.KW, d.stride, d.pad, d.groups = 3, 3, 1, 1, 1
        d.flags = L.RELU1 | ((L.SPLIT_OUT) if split_out is not None else 0)
        if split_out is not None:
            d.y_split = L.ptr(split_out)
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = 256, 256 * 2304, 256, 256
        d.tile, d.splitk, d.precision = 0, 1, prec
        import ctypes as C
        L.check(lib.vidc_conv2d_plan(C.byref(d)), "plan")
        d.splitk = 1
        L.check(lib.vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "conv")
    conv.y = torch.empty(4, 60, 80, 256, device=DEV)

    noises = {
        "none": None,
        "conv_fp32": lambda: conv(0),
        "conv_bf16x3": lambda: conv(1),
        "conv_fp32_splitout": lambda: conv(0, sp_a),
        "conv_bf16x3_splitout": lambda: conv(1, sp_a),
        "split": lambda: ops.split_bf16x3(xb),
        "stem_f32": lambda: ops.stem_conv3x3s2(ximg, wt),
        "stem_split": lambda: ops.stem_conv3x3s2(ximg, wt, split_out=sp_stem),
        "maxpool_f32": lambda: ops.maxpool3x3s2(xb),
        "maxpool_split": lambda: ops.maxpool3x3s2(xb, split_out=torch.empty(4, 60, 80, 256, device=DEV)),
        "wino_in": lambda: ops.winograd_input_transform(xa, 256, 4),
        "wino_in_split": lambda: ops.winograd_input_transform(xa, 256, 4, split=True),
        "memcpy": lambda: big1.copy_(big0, non_blocking=True),
        "fill": lambda: big1.mul_(1.0001),
    }
    nz = noises[a.noise]
    if nz is not None:
        with torch.cuda.stream(nst):
            nz()
        torch.cuda.synchronize()
    for it in range(a.iters):
        for k, v in enumerate(vs):
            v.go((it + 3 * k) % NI, ref)
        if nz is not None:
            with torch.cuda.stream(nst):
                for _ in range(a.per):
                    nz()
        if it % 16 == 15:       # bound the host's run-ahead
            vs[0].st.synchronize()
    torch.cuda.synchronize()
    counts = [int(c) for v in vs for c in v.bad]
    nbad = sum(1 for c in counts if c)
    print("NOISE %-22s victim=%s loads=%s victims=%d: %d of %d victim launches wrong (%d words)" % (
        a.noise, a.victim, os.environ.get("VIDC_DBG_STEM_LOADS", "0"), a.victims, nbad, len(counts), sum(counts)), flush=True)


if __name__ == "__main__":
    main()
