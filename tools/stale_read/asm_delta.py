"""Delta-debugging of hipcc's assembly of the fused stem: the victim kernel comes out of a CODE OBJECT assembled from a
(hand-edited) copy of the device assembly of csrc/pointwise.hip, the neighbour is the library's bf16x3 conv on another stream.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Ivi_depth_completion_amd/csrc -o pw.s vi_depth_completion_amd/csrc/pointwise.hip
    (edit pw.s)   clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c pw.s -o pw.o && ld.lld -shared pw.o -o pw.hsaco
    python tools/stale_read/asm_delta.py --hsaco pw.hsaco --loads 2 [--iters 1000] [--noise conv_bf16x3]

--loads N picks the template instantiation stem_conv_kernel<3, true, N> inside the code object (2 = vector record + plain tap loads: the failing form).
"""
import argparse
import ctypes as C
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
Ho, Wo = H // 2, W // 2


class Args(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int),
                ("ldy", C.c_int), ("relu", C.c_int), ("_pad0", C.c_int), ("ysp", C.c_void_p), ("ch0", C.c_int), ("_pad1", C.c_int), ("wp", C.c_void_p),
                ("wcx", C.c_float), ("wcy", C.c_float), ("ac", C.c_int)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hsaco", required=True)
    ap.add_argument("--loads", type=int, default=2)
    ap.add_argument("--noise", default="conv_bf16x3")
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--victims", type=int, default=2)
    ap.add_argument("--label", default="")
    ap.add_argument("--probe", action="store_true", help="selector weights (output channel k = patch element k) and a position-coded image: a wrong output IS the wrong patch value")
    a = ap.parse_args()
    assert C.sizeof(Args) == 96 and Args.ysp.offset == 56 and Args.wp.offset == 72 and Args.wcx.offset == 80
    lib = L.lib()
    torch.zeros(1, device=DEV)
    hip = C.CDLL("libamdhip64.so")
    mod, fn = C.c_void_p(), C.c_void_p()
    name = ("_ZN12_GLOBAL__N_116stem_conv_kernelILi3ELb1ELi%dEEEvPKfS2_PfiiiiiiiPtiS2_ffi" % a.loads).encode()
    rc = hip.hipModuleLoad(C.byref(mod), a.hsaco.encode())
    assert rc == 0, "hipModuleLoad -> %d" % rc
    rc = hip.hipModuleGetFunction(C.byref(fn), mod, name)
    assert rc == 0, "hipModuleGetFunction(%s) -> %d" % (name.decode(), rc)
    hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]

    NI = 8
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    imgs = [torch.rand((B, 3, H, W), device=DEV, generator=gen) for _ in range(NI)]
    gr = [torch.nn.functional.normalize(torch.tensor([[0.05 * (i - 3) + 0.01 * b, 1.0, 0.1 * (i - 4)] for b in range(B)]), dim=1).to(DEV) for i in range(NI)]
    al = torch.tensor([[0.0, 1.0, 0.0]] * B, device=DEV)
    kinv = torch.tensor(np.linalg.inv(np.array([[202.0, 0, 159.94], [0, 202.0, 119.94], [0, 0, 1.0]])).astype(np.float32).reshape(-1), device=DEV)
    wt = S.normal01(5, "stem.w", (64, 3, 3, 3), scale=0.2).float().to(DEV)
    if a.probe:
        wt = torch.zeros(64, 27)
        wt[torch.arange(27), torch.arange(27)] = 1.0
        wt = wt.view(64, 3, 3, 3).contiguous().to(DEV)
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        base = torch.stack([xx + 1000.0 * yy + 1.0e6 * c for c in range(3)])                    # (3, H, W): exact in fp32
        imgs = [torch.stack([base + 3.0e6 * ((i + b) % 4) for b in range(B)]).to(DEV) for i in range(NI)]
    saved = {"y": torch.zeros(B, Ho, Wo, 64, device=DEV), "ref": torch.zeros(B, Ho, Wo, 64, device=DEV), "n": torch.zeros((), dtype=torch.int64, device=DEV)}

    class Victim:
        def __init__(self):
            self.st = torch.cuda.Stream()
            self.x = torch.zeros(B, 3, H, W, device=DEV); self.x2 = torch.zeros_like(self.x); self.g = torch.zeros(B, 3, device=DEV)
            self.p = torch.zeros(B * 32, device=DEV); self.y = torch.zeros(B, Ho, Wo, 64, device=DEV)
            self.bad = []
            self.args = Args(self.x.data_ptr(), wt.data_ptr(), self.y.data_ptr(), H, W, Ho, Wo, 64, 64, 1, 0, None, 0, 0, self.p.data_ptr(), 159.94, 119.94, 0)
            self.size = C.c_size_t(92)
            self.extra = (C.c_void_p * 5)(1, C.cast(C.pointer(self.args), C.c_void_p), 2, C.cast(C.pointer(self.size), C.c_void_p), 3)

        def go(self, i, ref=None, use_lib=False):
            with torch.cuda.stream(self.st):
                s = self.st.cuda_stream
                self.x2.copy_(self.x, non_blocking=True)
                for b in range(B):
                    self.x[b:b + 1].copy_(imgs[i][b:b + 1], non_blocking=True)
                self.g.copy_(gr[i], non_blocking=True)
                L.check(lib.vidc_warp2dof_params(self.g.data_ptr(), al.data_ptr(), B, 202.0, 202.0, 159.94, 119.94, kinv.data_ptr(), W, H, self.p.data_ptr(), s), "p")
                if use_lib:
                    L.check(lib.vidc_stem_conv3x3s2_warped(self.x.data_ptr(), self.p.data_ptr(), wt.data_ptr(), self.y.data_ptr(), B, H, W, 64, 64, 1, None, 0, 159.94, 119.94, 0, s), "stem")
                else:
                    rc = hip.hipModuleLaunchKernel(fn, (Wo + 63) // 64, Ho, B, 256, 1, 1, 0, s, None, C.cast(self.extra, C.c_void_p))
                    assert rc == 0, "hipModuleLaunchKernel -> %d" % rc
                if ref is not None:
                    nb = (self.y.view(torch.int32) != ref[i].view(torch.int32)).sum()
                    self.bad.append(nb)
                    if a.probe:                   # keep the last wrong output and what it should have been
                        hit = nb > 0
                        saved["y"] = torch.where(hit, self.y, saved["y"])
                        saved["ref"] = torch.where(hit, ref[i], saved["ref"])
                        saved["n"] = saved["n"] + hit.to(torch.int64)
                    return None
                return self.y.clone()

    vs = [Victim() for _ in range(a.victims)]
    torch.cuda.synchronize()
    ref = []
    for i in range(NI):
        ref.append(vs[0].go(i, use_lib=True)); torch.cuda.synchronize()       # the library's shipped form, quiescent
    chk = vs[0].go(0); torch.cuda.synchronize()
    assert torch.equal(chk, ref[0]), "the code object's kernel does not reproduce the library's result on a quiet device"

    nst = torch.cuda.Stream()
    xa = torch.randn(4, 60, 80, 256, device=DEV)
    w33 = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
    one, zero = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    wpk = {0: ops.pack_conv_weight(w33), 1: ops.pack_conv_weight_bf16x3(w33)}
    xin = {0: xa, 1: ops.split_bf16x3(xa)}
    yv = torch.empty(4, 60, 80, 256, device=DEV)

    def conv(prec):
        d = L.ConvDesc()
        d.x, d.w, d.y = L.ptr(xin[prec]), L.ptr(wpk[prec]), L.ptr(yv)
        d.scale1, d.shift1 = L.ptr(one), L.ptr(zero)
        d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 4, 60, 80, 256, 256, 60, 80, 256, 256
        d.KH, d.KW, d.stride, d.pad, d.groups, d.flags = 3, 3, 1, 1, 1, L.RELU1
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = 256, 256 * 2304, 256, 256
        d.tile, d.splitk, d.precision = 0, 1, prec
        L.check(lib.vidc_conv2d_plan(C.byref(d)), "plan")
        d.splitk = 1
        L.check(lib.vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "conv")

    nz = {"none": None, "conv_fp32": lambda: conv(0), "conv_bf16x3": lambda: conv(1)}[a.noise]
    for it in range(a.iters):
        for k, v in enumerate(vs):
            v.go((it + 3 * k) % NI, ref)
        if nz is not None:
            with torch.cuda.stream(nst):
                nz(); nz()
        if it % 16 == 15:
            vs[0].st.synchronize()
    torch.cuda.synchronize()
    counts = [int(c) for v in vs for c in v.bad]
    nbad = sum(1 for c in counts if c)
    print("ASM %-40s loads=%d noise=%s: %d of %d victim launches wrong (%d words)" % (a.label or os.path.basename(a.hsaco), a.loads, a.noise, nbad, len(counts), sum(counts)), flush=True)
    if a.probe and int(saved["n"]) > 0:
        y, r = saved["y"].cpu(), saved["ref"].cpu()
        ne = torch.nonzero(y != r)
        print("  last wrong launch: %d wrong words; (b, oy, ox, k = c*9 + r*3 + s): expected -> got" % ne.shape[0])
        seen = 0
        for b, oy, ox, k in ne.tolist():
            if k >= 27:
                continue
            c, rr, ss = k // 9, (k // 3) % 3, k % 3
            print("    b %d oy %3d ox %3d  c %d r %d s %d (patch col %3d): %14.3f -> %14.3f   (diff %+.3f)" % (b, oy, ox, c, rr, ss, 2 * (ox % 64) + ss, float(r[b, oy, ox, k]), float(y[b, oy, ox, k]),
                                                                                                        float(y[b, oy, ox, k]) - float(r[b, oy, ox, k])))
            seen += 1
            if seen >= 60:
                break


if __name__ == "__main__":
    main()
