"""Functional (tensor in -> tensor out) wrappers over single libvidc.so entry points.

The networks do not go through these (they run as whole `engine.Program`s); they exist so that every C entry
point can be exercised and parity-tested on its own, and for the plane block host code.  All tensors must be
CUDA/HIP tensors; nothing here computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib as L


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vidc ops take GPU tensors only (no CPU fallback)")


def pack_conv_weight(w_oihw):
    _dev(w_oihw)
    w = w_oihw.contiguous().float()
    co, ci, kh, kw = w.shape
    out = torch.empty((co, kh * kw * ci), dtype=torch.float32, device=w.device)
    L.check(L.lib().vidc_pack_conv_weight(L.ptr(w), L.ptr(out), co, ci, kh, kw, L.current_stream()), "pack_conv_weight")
    return out


def pack_conv_weight_bf16x3(w_oihw):
    """OIHW fp32 -> split-bf16 packed weights; returned as a float32-typed (Cout, K) tensor (same bytes: each 32-wide
    K unit is [32 x bf16 hi | 32 x bf16 lo])."""
    _dev(w_oihw)
    w = w_oihw.contiguous().float()
    co, ci, kh, kw = w.shape
    out = torch.empty((co, kh * kw * ci), dtype=torch.float32, device=w.device)
    L.check(L.lib().vidc_pack_conv_weight_bf16x3(L.ptr(w), L.ptr(out), co, ci, kh, kw, L.current_stream()), "pack_conv_weight_bf16x3")
    return out


def split_bf16x3(x_nhwc):
    """fp32 NHWC -> split-bf16 image of the same shape/bytes (float32-typed storage)."""
    _dev(x_nhwc)
    x = x_nhwc.contiguous().float()
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    L.check(L.lib().vidc_split_bf16x3(L.ptr(x), L.ptr(out), x.numel() // Cc, Cc, Cc, L.current_stream()), "split_bf16x3")
    return out


def conv2d_bn_act(x, w_packed, scale1, shift1, kh, kw, stride=1, pad=0, relu1=False, scale2=None, shift2=None, relu2=False,
                  residual=None, relu3=False, accumulate_into=None, tile=0, splitk=1, groups=1, precision=0, split_out=None,
                  no_f32_out=False, workspace=None):
    """x: NHWC (B,H,W,G*Cin) contiguous; w_packed: (G,Cout,kh*kw*Cin) or (Cout,K); returns NHWC (B,Ho,Wo,G*Cout).
    precision=1 (bf16x3): x is split here; w_packed must come from pack_conv_weight_bf16x3.
    workspace: optional persistent split-K scratch (float32, zero-initialised once by the caller; include/vidc.h)."""
    _dev(x, w_packed, scale1, shift1)
    x = x.contiguous()
    if precision == L.PREC_BF16X3:
        x = split_bf16x3(x)
    B, H, W, ld = x.shape
    G = groups
    cin = ld // G
    wp = w_packed.contiguous().view(G, -1, kh * kw * cin)
    cout = wp.shape[1]
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = accumulate_into if accumulate_into is not None else torch.empty((B, Ho, Wo, G * cout), dtype=torch.float32, device=x.device)
    d = L.ConvDesc()
    d.x, d.w, d.y = L.ptr(x), L.ptr(wp), L.ptr(y)
    s1, b1 = scale1.contiguous().float(), shift1.contiguous().float()
    d.scale1, d.shift1 = L.ptr(s1), L.ptr(b1)
    flags = (L.RELU1 if relu1 else 0)
    keep = [s1, b1]
    if scale2 is not None:
        s2, b2 = scale2.contiguous().float(), shift2.contiguous().float()
        d.scale2, d.shift2 = L.ptr(s2), L.ptr(b2)
        keep += [s2, b2]
        flags |= L.AFFINE2 | (L.RELU2 if relu2 else 0)
    if residual is not None:
        residual = residual.contiguous()
        d.residual, d.ldr, d.r_gs = L.ptr(residual), residual.shape[-1], cout
        flags |= L.RESIDUAL | (L.RELU3 if relu3 else 0)
    if accumulate_into is not None:
        flags |= L.ACCUM
    if split_out is not None:       # float32-typed tensor of y's shape receiving the split-bf16 image of the result
        d.y_split = L.ptr(split_out)
        flags |= L.SPLIT_OUT | (L.NO_F32_OUT if no_f32_out else 0)
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, cin, ld
    d.Ho, d.Wo, d.Cout, d.ldy = Ho, Wo, cout, G * cout
    d.KH, d.KW, d.stride, d.pad, d.flags, d.groups = kh, kw, stride, pad, flags, G
    d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, cout * kh * kw * cin, cout, (cout if s1.numel() >= G * cout else 0)      # (one affine for all groups: p_gs = 0)
    d.tile, d.splitk, d.precision = tile, splitk, precision
    if tile == 0:
        L.check(L.lib().vidc_conv2d_plan(C.byref(d)), "conv2d_plan")
        if splitk > 1:
            d.splitk = splitk
    ws = None
    nbytes = L.lib().vidc_conv2d_workspace_bytes(C.byref(d))
    if nbytes and workspace is not None:
        if workspace.numel() * 4 < nbytes:
            raise RuntimeError("split-K workspace too small: %d < %d bytes" % (workspace.numel() * 4, nbytes))
        d.workspace = L.ptr(workspace)
    elif nbytes:
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=x.device)     # zeroed: ticket counters at its head
        d.workspace = L.ptr(ws)
    L.check(L.lib().vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "conv2d_bn_act")
    return y


def stem_conv3x3s2(x_nchw, w_oihw, relu=True, split_out=None):
    """split_out: optional float32-typed tensor of y's shape receiving the split-bf16 image of the result (Cout % 32 == 0)."""
    _dev(x_nchw, w_oihw)
    x, w = x_nchw.contiguous().float(), w_oihw.contiguous().float()
    B, cin, H, W = x.shape
    co = w.shape[0]
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((B, Ho, Wo, co), dtype=torch.float32, device=x.device)
    L.check(L.lib().vidc_stem_conv3x3s2(L.ptr(x), L.ptr(w), L.ptr(y), B, cin, H, W, co, co, int(relu), L.ptr(split_out), 0,
                                        L.current_stream()), "stem")
    return y


def stem_conv3x3s2_warped(x_nchw, warp_params, w_oihw, cx, cy, align_corners=False, relu=True):
    """The stem conv on the gravity-aligned forward warp of x, gathered on the fly (the warped image is never stored)."""
    _dev(x_nchw, warp_params, w_oihw)
    x, w = x_nchw.contiguous().float(), w_oihw.contiguous().float()
    B, cin, H, W = x.shape
    assert cin == 3
    co = w.shape[0]
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((B, Ho, Wo, co), dtype=torch.float32, device=x.device)
    L.check(L.lib().vidc_stem_conv3x3s2_warped(L.ptr(x), L.ptr(warp_params), L.ptr(w), L.ptr(y), B, H, W, co, co, int(relu), None, 0, float(cx), float(cy),
                                               int(align_corners), L.current_stream()), "stem (warped input)")
    return y


def maxpool3x3s2(x, split_out=None):
    _dev(x)
    x = x.contiguous()
    B, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((B, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
    L.check(L.lib().vidc_maxpool3x3s2(L.ptr(x), L.ptr(y), B, H, W, Cc, Cc, Cc, L.ptr(split_out), L.current_stream()), "maxpool")
    return y


def upsample_bilinear_ac(x, size, relu=False, accumulate_into=None, split_out=None, store_f32=True, sum_groups=1):
    """sum_groups = G > 1: x holds G groups of C/G channels; their upsampled (ReLU'd) values are added into accumulate_into (C/G ch)."""
    _dev(x)
    x = x.contiguous()
    B, h, w, Cc = x.shape
    if sum_groups > 1:
        assert accumulate_into is not None and Cc % sum_groups == 0
        y = accumulate_into
        co = Cc // sum_groups
        flags = (L.UP_RELU if relu else 0) | L.UP_ACCUM | (sum_groups << 8)
        L.check(L.lib().vidc_upsample_bilinear_ac(L.ptr(x), L.ptr(y), B, h, w, co, Cc, size[0], size[1], co, flags, L.ptr(split_out),
                                                  L.current_stream()), "upsample")
        return y
    y = accumulate_into if accumulate_into is not None else torch.empty((B, size[0], size[1], Cc), dtype=torch.float32, device=x.device)
    flags = (L.UP_RELU if relu else 0) | (L.UP_ACCUM if accumulate_into is not None else 0) | (0 if store_f32 else L.UP_NO_F32_OUT)
    L.check(L.lib().vidc_upsample_bilinear_ac(L.ptr(x), L.ptr(y), B, h, w, Cc, Cc, size[0], size[1], Cc, flags, L.ptr(split_out),
                                              L.current_stream()), "upsample")
    return y


def head_conv1x1_upsample(x, w, bias, pad, size, relu):
    """x NHWC (B,h,w,Cin); w (Cout,Cin[,1,1]); returns (y NCHW (B,Cout,H,W), lowres NCHW (B,Cout,h+2p,w+2p))."""
    _dev(x, w, bias)
    x = x.contiguous()
    B, h, wd, cin = x.shape
    w2 = w.reshape(w.shape[0], -1).contiguous().float()
    co = w2.shape[0]
    low = torch.empty((B, co, h + 2 * pad, wd + 2 * pad), dtype=torch.float32, device=x.device)
    y = torch.empty((B, co, size[0], size[1]), dtype=torch.float32, device=x.device)
    b = bias.contiguous().float()
    L.check(L.lib().vidc_head_conv1x1_upsample(L.ptr(x), L.ptr(w2), L.ptr(b), L.ptr(low), L.ptr(y), B, h, wd, cin, cin, co, pad,
                                               size[0], size[1], int(relu), L.current_stream()), "head")
    return y, low


# ---- Winograd F(m x m, 3x3) (csrc/winograd.hip) ------------------------------------------------------------------------------
def winograd_weight_transform(w_oihw, m):
    """OIHW (Cout,Cin,3,3) -> U (a*a, Cout, Cin), a = m + 2."""
    _dev(w_oihw)
    w = w_oihw.contiguous().float()
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    a2 = (m + 2) * (m + 2)
    u = torch.empty((a2, co, ci), dtype=torch.float32, device=w.device)
    L.check(L.lib().vidc_winograd_weight_transform(L.ptr(w), L.ptr(u), co, ci, m, L.current_stream()), "winograd_weight_transform")
    return u


def winograd_input_transform(x, cin, m, split=False):
    """x NHWC (B,H,W,G*cin) -> V (tiles, G*a*a*cin) [tile][gg][pos][cin]; split=True: the split-bf16 image (float32-typed storage)."""
    _dev(x)
    x = x.contiguous()
    B, H, W, Cc = x.shape
    th, tw = -(-H // m), -(-W // m)
    a2 = (m + 2) * (m + 2)
    v = torch.empty((B * th * tw, a2 * Cc), dtype=torch.float32, device=x.device)
    L.check(L.lib().vidc_winograd_input_transform(L.ptr(x), L.ptr(v), B, H, W, Cc, Cc, cin, m, int(split), 0, L.current_stream()),
            "winograd_input_transform")
    return v


def winograd_output_transform(mm, B, Ho, Wo, cout, m, scale1, shift1, relu1=False, scale2=None, shift2=None, relu2=False, split_out=None,
                              no_f32_out=False):
    """mm (tiles, G*a*a*cout) -> y NHWC (B,Ho,Wo,G*cout) = epilogue(A^T M A)."""
    _dev(mm, scale1, shift1)
    mm = mm.contiguous()
    a2 = (m + 2) * (m + 2)
    Cc = mm.shape[1] // a2
    y = torch.empty((B, Ho, Wo, Cc), dtype=torch.float32, device=mm.device)
    s1, b1 = scale1.contiguous().float().view(-1), shift1.contiguous().float().view(-1)
    flags = (L.RELU1 if relu1 else 0)
    s2 = b2 = None
    if scale2 is not None:
        s2, b2 = scale2.contiguous().float().view(-1), shift2.contiguous().float().view(-1)
        flags |= L.AFFINE2 | (L.RELU2 if relu2 else 0)
    if split_out is not None:
        flags |= L.SPLIT_OUT | (L.NO_F32_OUT if no_f32_out else 0)
    L.check(L.lib().vidc_winograd_output_transform(L.ptr(mm), L.ptr(y), L.ptr(split_out), L.ptr(s1), L.ptr(b1), L.ptr(s2), L.ptr(b2), B, Ho, Wo,
                                                   Cc, cout, Cc, m, flags, 0, L.current_stream()), "winograd_output_transform")
    return y


def conv3x3_winograd(x, w_oihw_groups, scale1, shift1, m, relu1=False, scale2=None, shift2=None, relu2=False, precision=0, tile=0, splitk=1,
                     split_out=None, no_f32_out=False):
    """nn.Conv2d(cin, cout, 3, 1, 1) [+ per-channel affines / ReLUs] of G groups as Winograd F(m x m, 3x3): input transform, ONE
    grouped 1x1 GEMM launch (a*a*G groups, identity epilogue) on the MFMA kernel, output transform with the epilogue.
    x NHWC (B,H,W,G*cin); w_oihw_groups: list of G (cout,cin,3,3) tensors; scale / shift: (G, cout)."""
    G = len(w_oihw_groups)
    B, H, W, Cc = x.shape
    cin = Cc // G
    cout = w_oihw_groups[0].shape[0]
    a2 = (m + 2) * (m + 2)
    u = torch.cat([winograd_weight_transform(w, m) for w in w_oihw_groups], 0)              # (G*a2, cout, cin): group gg*a2 + pos
    if precision == L.PREC_BF16X3:
        img = torch.empty_like(u)
        L.check(L.lib().vidc_pack_conv_weight_bf16x3(L.ptr(u), L.ptr(img), G * a2 * cout, cin, 1, 1, L.current_stream()), "pack")
        u = img
    v = winograd_input_transform(x, cin, m, split=precision == L.PREC_BF16X3)
    tiles = v.shape[0]
    d = L.ConvDesc()
    mm = torch.empty((tiles, a2 * G * cout), dtype=torch.float32, device=x.device)
    one, zero = torch.ones(cout, dtype=torch.float32, device=x.device), torch.zeros(cout, dtype=torch.float32, device=x.device)
    d.x, d.w, d.y, d.scale1, d.shift1 = L.ptr(v), L.ptr(u), L.ptr(mm), L.ptr(one), L.ptr(zero)
    d.B, d.H, d.W, d.Cin, d.ldx = 1, 1, tiles, cin, a2 * G * cin
    d.Ho, d.Wo, d.Cout, d.ldy = 1, tiles, cout, a2 * G * cout
    d.KH, d.KW, d.stride, d.pad, d.flags, d.groups = 1, 1, 1, 0, 0, a2 * G
    d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, cout * cin, cout, 0
    d.tile, d.splitk, d.precision = tile, splitk, precision
    if tile == 0:
        L.check(L.lib().vidc_conv2d_plan(C.byref(d)), "conv2d_plan")
        if splitk > 1:
            d.splitk = splitk
    ws = None
    nbytes = L.lib().vidc_conv2d_workspace_bytes(C.byref(d))
    if nbytes:
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=x.device)
        d.workspace = L.ptr(ws)
    L.check(L.lib().vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "conv2d_bn_act (winograd GEMMs)")
    return winograd_output_transform(mm, B, H, W, cout, m, scale1, shift1, relu1, scale2, shift2, relu2, split_out, no_f32_out)


def winograd_weight_pack_fused(u):
    """U (36, Cout, Cin) of winograd_weight_transform(w, 4) -> the fragment order the fused kernel streams (same shape / bytes, another order)."""
    _dev(u)
    u = u.contiguous().float()
    a2, co, ci = u.shape
    assert a2 == 36
    out = torch.empty_like(u)
    L.check(L.lib().vidc_winograd_weight_pack_fused(L.ptr(u), L.ptr(out), co, ci, L.current_stream()), "winograd_weight_pack_fused")
    return out


def conv3x3_winograd_fused(x, w_oihw_groups, scale1, shift1, relu1=False, scale2=None, shift2=None, relu2=False, u=None):
    """The same layer as conv3x3_winograd(m = 4) in ONE launch (csrc/wfused.hip, tile VIDC_TILE_WINO4_FUSED): input transform, the 36 products
    and the output transform + epilogue per block of 16 tiles x 32 output channels; no V / M tensors.  fp32 only.
    x NHWC (B,H,W,G*cin); w_oihw_groups: list of G (cout,cin,3,3) tensors (or `u`: their transformed AND packed weights, (G*36,cout,cin) floats); scale / shift: (G, cout)."""
    _dev(x, scale1, shift1)
    x = x.contiguous()
    G = len(w_oihw_groups) if u is None else u.shape[0] // 36
    B, H, W, Cc = x.shape
    cin = Cc // G
    if u is None:
        u = torch.cat([winograd_weight_pack_fused(winograd_weight_transform(w, 4)) for w in w_oihw_groups], 0)      # (G*36, cout, cin) floats, fragment order
    cout = u.shape[1]
    y = torch.empty((B, H, W, G * cout), dtype=torch.float32, device=x.device)
    d = L.ConvDesc()
    s1, b1 = scale1.contiguous().float().view(-1), shift1.contiguous().float().view(-1)
    d.x, d.w, d.y, d.scale1, d.shift1 = L.ptr(x), L.ptr(u), L.ptr(y), L.ptr(s1), L.ptr(b1)
    flags = (L.RELU1 if relu1 else 0)
    keep = [s1, b1]
    if scale2 is not None:
        s2, b2 = scale2.contiguous().float().view(-1), shift2.contiguous().float().view(-1)
        d.scale2, d.shift2 = L.ptr(s2), L.ptr(b2)
        keep += [s2, b2]
        flags |= L.AFFINE2 | (L.RELU2 if relu2 else 0)
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, cin, Cc
    d.Ho, d.Wo, d.Cout, d.ldy = H, W, cout, G * cout
    d.KH, d.KW, d.stride, d.pad, d.flags, d.groups = 3, 3, 1, 1, flags, G
    d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, 36 * cout * cin, cout, (cout if s1.numel() >= G * cout else 0)
    d.tile, d.splitk, d.precision = L.TILE_WINO4_FUSED, 1, L.PREC_FP32
    L.check(L.lib().vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "conv2d_bn_act (fused Winograd)")
    return y


def clock_stamps(n, device=None):
    """Buffer for n stamps of `clock_stamp`."""
    return torch.zeros((n, L.CLOCK_STAMP_WGS, 4), dtype=torch.int64, device=device if device is not None else "cuda")


def clock_stamp(stamps, i):
    """Enqueues stamp i on the current stream (include/vidc.h vidc_clock_stamp): per XCD the shader-clock cycle counter and the 100 MHz wall
    clock.  `shader_clock_ghz(stamps, a, b)` turns two stamps into the average shader clock between them."""
    if not stamps.is_cuda or stamps.dtype != torch.int64 or tuple(stamps.shape[1:]) != (L.CLOCK_STAMP_WGS, 4) or not stamps.is_contiguous():
        raise RuntimeError("clock_stamp: a buffer made by ops.clock_stamps is required")
    L.check(L.lib().vidc_clock_stamp(stamps.data_ptr() + 8 * 4 * L.CLOCK_STAMP_WGS * int(i), L.current_stream()), "clock_stamp")


def shader_clock_ghz(stamps, a, b):
    """Median over the XCDs present in both stamps of (cycle difference) / (100 MHz tick difference) x 0.1; None if they share no XCD."""
    h = stamps.cpu()
    sa = {int(r[0]): (int(r[1]), int(r[2])) for r in h[a] if int(r[3]) == 1}
    sb = {int(r[0]): (int(r[1]), int(r[2])) for r in h[b] if int(r[3]) == 1}
    ghz = sorted((sb[x][0] - sa[x][0]) / (sb[x][1] - sa[x][1]) * 0.1 for x in sa if x in sb and sb[x][1] > sa[x][1] and sb[x][0] > sa[x][0])
    return ghz[len(ghz) // 2] if ghz else None
