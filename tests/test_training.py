"""Training step of the depth-completion network (BASELINE configs[4]).

CPU: gradient bucketing + the cross-rank SUM on two gloo ranks.  GPU: every backward kernel against torch-CPU autograd on small
layers, then ONE whole `_run_training_iteration` against the fixture produced by the reference itself
(tests/golden/train_step.npz, oracle/tools/make_golden_train.py): loss, 29 gradient tensors spread over the network, the parameters
after the Adam step, updated running statistics.  Tolerances are stated per test (fp32 everywhere; reductions in fp64 on both sides)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from vi_depth_completion_amd import synthetic as S

gpu = pytest.mark.gpu
DEV = "cuda"


def test_gradient_buckets_cover_the_buffer():
    from vi_depth_completion_amd.training import GradientBuckets
    b = GradientBuckets(100, 32)
    assert b.ranges == [(0, 32), (32, 64), (64, 96), (96, 100)]
    assert GradientBuckets(64, 32).ranges == [(0, 32), (32, 64)]
    flat = torch.arange(10.0)
    assert b.all_reduce(flat) is flat            # no process group: untouched


def _bucket_worker(rank, world, port, out):
    import torch.distributed as dist
    from vi_depth_completion_amd.training import GradientBuckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    GradientBuckets(1000, 256).all_reduce(flat)
    # the split the trainer uses across ranks: the tail of the buffer (the decoder's gradients) first, the head later, each in its
    # own buckets; callbacks called before the data is used
    two = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    b = GradientBuckets(1000, 256)
    waits = b.all_reduce_async(two, 600, None)
    waits += b.all_reduce_async(two, 0, 600)
    for w in waits:
        w()
    assert torch.equal(two, flat)
    if rank == 0:
        out.put(flat.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_gradients_are_summed_over_two_ranks_gloo():
    """Frames shard over ranks and the loss is a SUM over the whole batch divided by a constant (network_run.py:173), so the whole-batch
    gradient is the sum of the ranks' gradients: bucketed all_reduce(SUM) on two gloo ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, 29517, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(got, torch.arange(1000, dtype=torch.float32) * 3)


# ---- GPU: kernels vs torch autograd ------------------------------------------------------------------------------------------------
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _trainer(module):
    """Kernel-level comparisons run with exact fp32 products: in the default bf16x3 mode the forward differs by ~1e-5, which flips the
    ReLU gate of the odd pre-activation that sits within 1e-5 of zero, and ONE flipped gate moves max|dx| by percents of the scale
    (measured: 1.6 %) although every kernel is right (dgrad alone: 1.4e-5).  The whole-network test covers both modes."""
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    assert os.environ.get("VIDC_TRAIN_PRECISION", "fp32") == "fp32"
    tr = DepthCompletionTrainer(module.to(DEV), 1e-3)
    tr.tape = []
    return tr


def _run_tape(tr):
    for fn in reversed(tr.tape):
        fn()
    tr.tape = []
    torch.cuda.synchronize()


def _close(a, b, rtol, name):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-12
    assert err <= rtol * scale, "%s: max |diff| %.3e vs scale %.3e" % (name, err, scale)


@gpu
@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W", [(64, 96, 3, 1, 1, 12, 20), (64, 64, 3, 2, 1, 15, 20), (128, 256, 1, 2, 0, 15, 20), (96, 32, 1, 1, 0, 8, 10)])
def test_conv_bn_relu_block_gradients(cin, cout, k, stride, pad, H, W):
    """conv(+bias) -> BatchNorm(train) -> ReLU: forward, running statistics and all five gradients (x, W, b, gamma, beta) vs torch-CPU
    autograd; covers the strided dgrad (zero-stuffed dY) and the 1x1 / 3x3 wgrad.  2e-4 of the tensor's scale (fp32 sums of up to ~10^4
    terms in another order; BatchNorm divides by sqrt(var))."""
    from vi_depth_completion_amd.training import Act
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Conv2d(cin, cout, k, stride, pad), nn.BatchNorm2d(cout)).train()
    with torch.no_grad():
        ref[1].weight.uniform_(0.5, 1.5)
        ref[1].bias.normal_(0, 0.2)
    import copy
    mod = copy.deepcopy(ref)
    x = torch.randn(3, cin, H, W)
    gw = torch.randn(3, cout, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        yr = F.relu(ref(xr))
        (yr * gw).sum().backward()
    tr = _trainer(mod)
    with torch.no_grad():
        xa = Act(_nhwc(x).to(DEV))
        y = tr.bn(tr.conv(xa, "0", stride, pad), "1", True)
        tr.flush_counters()
        y.grad = _nhwc(gw).to(DEV)
        _run_tape(tr)
    _close(_nchw(y.t), yr, 2e-5, "forward")
    _close(_nchw(xa.grad), xr.grad, 2e-4, "dx")
    _close(tr.grad["0.weight"], ref[0].weight.grad, 2e-4, "dW")
    # a bias in front of a BatchNorm has gradient 0 in exact arithmetic: both sides hold rounding noise of sums of ~10^3 O(1) terms
    assert float(tr.grad["0.bias"].abs().max()) < 2e-4 and float(ref[0].bias.grad.abs().max()) < 2e-4
    _close(tr.grad["1.weight"], ref[1].weight.grad, 2e-4, "dgamma")
    _close(tr.grad["1.bias"], ref[1].bias.grad, 2e-4, "dbeta")
    _close(mod[1].running_mean, ref[1].running_mean, 1e-5, "running_mean")
    _close(mod[1].running_var, ref[1].running_var, 1e-5, "running_var")
    assert int(mod[1].num_batches_tracked) == 1


@gpu
def test_maxpool_backward_wide_kernel_equals_scalar_and_autograd():
    """vidc_maxpool3x3s2_backward: the 4-channels-per-thread kernel (aligned, channel strides multiples of 4) against the scalar one (forced
    by an odd gradient stride) -- identical bits -- and against torch autograd on the CPU, ties included (first maximum in scan order)."""
    from vi_depth_completion_amd import _lib as L
    lib = L.lib()
    B, H, W, Cc = 2, 11, 14, 8
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, 4, (B, H, W, Cc), generator=g).float()            # many ties
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Ho, Wo, Cc, generator=g)
    with torch.enable_grad():          # (other tests of the suite switch gradient recording off globally)
        xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
        F.max_pool2d(xr, 3, 2, 1).backward(dy.permute(0, 3, 1, 2))
    want = xr.grad.permute(0, 2, 3, 1)
    xd, dyd = x.to(DEV), dy.to(DEV)
    wide = torch.empty(B, H, W, Cc, device=DEV)
    L.check(lib.vidc_maxpool3x3s2_backward(L.ptr(xd), L.ptr(dyd), L.ptr(wide), B, H, W, Cc, Cc, Cc, Cc, L.current_stream()), "maxpool_bwd")
    odd = torch.empty(B, H, W, Cc + 1, device=DEV)
    L.check(lib.vidc_maxpool3x3s2_backward(L.ptr(xd), L.ptr(dyd), L.ptr(odd), B, H, W, Cc, Cc, Cc, Cc + 1, L.current_stream()), "maxpool_bwd scalar")
    assert torch.equal(wide.cpu(), odd[..., :Cc].cpu())
    assert torch.equal(wide.cpu(), want)
    # bilinear upsampling (align_corners=True) backward: the 4-channel kernel against the scalar one, and against autograd
    h, w, Hh, Ww = 8, 10, 15, 20
    gy = torch.randn(B, Hh, Ww, Cc, generator=g)
    with torch.enable_grad():
        src = torch.randn(B, Cc, h, w, generator=g).requires_grad_(True)
        F.interpolate(src, size=(Hh, Ww), mode="bilinear", align_corners=True).backward(gy.permute(0, 3, 1, 2))
    gyd = gy.to(DEV)
    wide = torch.empty(B, h, w, Cc, device=DEV)
    L.check(lib.vidc_upsample_bilinear_ac_backward(L.ptr(gyd), L.ptr(wide), B, h, w, Cc, Cc, Cc, Hh, Ww, L.current_stream()), "upsample_bwd")
    odd = torch.empty(B, h, w, Cc + 1, device=DEV)
    L.check(lib.vidc_upsample_bilinear_ac_backward(L.ptr(gyd), L.ptr(odd), B, h, w, Cc, Cc, Cc + 1, Hh, Ww, L.current_stream()), "upsample_bwd scalar")
    assert torch.equal(wide.cpu(), odd[..., :Cc].cpu())
    assert (wide.cpu() - src.grad.permute(0, 2, 3, 1)).abs().max() < 1e-5


@gpu
def test_bottleneck_pool_upsample_gradients():
    """A miniature of the network's graph: max-pool -> projection Bottleneck (stride 2) -> identity Bottleneck -> upsample -> add, so the
    fan-outs (block input feeds conv1 AND the shortcut) accumulate; gradients w.r.t. the input and a few parameters vs torch."""
    from vi_depth_completion_amd.networks.backbone import Bottleneck
    from vi_depth_completion_amd.training import Act
    torch.manual_seed(1)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.b0 = Bottleneck(64, 32, 2, True)
            self.b1 = Bottleneck(128, 32, 1, False)

        def block(self, blk, x):
            t = F.relu(blk.bn1(blk.conv1(x)))
            t = F.relu(blk.bn2(blk.conv2(t)))
            t = blk.bn3(blk.conv3(t))
            idn = x if blk.downsample is None else blk.downsample(x)
            return F.relu(t + idn)

        def forward(self, x):
            p = F.max_pool2d(x, 3, 2, 1)
            a = self.block(self.b0, p)
            b = self.block(self.b1, a)
            u = F.interpolate(b, size=(p.shape[2], p.shape[3]), mode="bilinear", align_corners=True)
            return u[:, :64] + p

    ref = Net().train()
    import copy
    mod = copy.deepcopy(ref)
    x = torch.randn(2, 64, 30, 41)
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        yr = ref(xr)
        gw = torch.randn_like(yr)
        (yr * gw).sum().backward()
    tr = _trainer(mod)
    with torch.no_grad():
        xa = Act(_nhwc(x).to(DEV))
        p = tr.maxpool(xa)
        a = tr._bottleneck(p, "b0.", 2, True)
        b = tr._bottleneck(a, "b1.", 1, False)
        u = tr.upsample(b, (p.t.shape[1], p.t.shape[2]))
        u64 = Act(u.t[..., :64])
        # the slice's gradient is a slice of u's gradient: pre-allocate u.grad (zeros beyond channel 64)
        u.grad = torch.zeros_like(u.t)
        u64.grad = None
        y = tr.add(u64, p, False)
        y.grad = _nhwc(gw).to(DEV)
        tr.tape[-1]()                                 # backward of the add: u64.grad and p.grad (+=) are set
        u.grad[..., :64].copy_(u64.grad)
        tr.tape.pop()
        _run_tape(tr)
    _close(_nchw(y.t), yr, 2e-5, "forward")
    _close(_nchw(xa.grad), xr.grad, 3e-4, "dx")
    for k in ("b0.conv2.weight", "b0.downsample.0.weight", "b0.bn3.weight", "b1.conv1.weight", "b1.bn2.bias", "b1.conv3.weight"):
        rp = dict(ref.named_parameters())[k]
        _close(tr.grad[k], rp.grad, 3e-4, k)


@gpu
def test_loss_and_adam_kernels():
    from vi_depth_completion_amd import _lib as L
    torch.manual_seed(2)
    pred = torch.rand(2, 1, 24, 32) * 4
    gt = torch.rand(2, 1, 24, 32) * 4
    gt[gt < 0.8] = 0.0
    pred[0, 0, 0, 0] = gt[0, 0, 0, 0] = 2.5            # exact tie: sign(0) = 0
    pr = pred.clone().requires_grad_(True)
    with torch.enable_grad():
        m = gt > 0
        lr = F.l1_loss(pr[m], gt[m], reduction="sum") / (24 * 32)
        lr.backward()
    n = pred.numel()
    loss = torch.zeros((), dtype=torch.float64, device=DEV)
    dp, terms = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    sc = torch.empty(4096, dtype=torch.uint8, device=DEV)
    pd, gd = pred.to(DEV), gt.to(DEV)
    L.check(L.lib().vidc_masked_l1_loss(L.ptr(pd), L.ptr(gd), n, 24 * 32, L.ptr(loss), L.ptr(dp), L.ptr(terms), L.ptr(sc), L.current_stream()), "loss")
    assert abs(float(loss) - float(lr.detach())) < 1e-6 * float(lr.detach())
    assert torch.equal(dp.cpu().view_as(pred), pr.grad)
    # Adam: three steps of torch.optim.Adam on the same gradients
    p0 = torch.randn(1000)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-2)
    p, mm, vv = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    # (a second copy of 1003 values whose last three repeat the first three: the scalar tail of the launch must round like its four-wide body)
    rep = lambda t: torch.cat([t, t[:3]])      # noqa: E731
    p3, m3, v3 = rep(p0).to(DEV), torch.zeros(1003, device=DEV), torch.zeros(1003, device=DEV)
    for step in (1, 2, 3):
        g = torch.randn(1000) * (10.0 ** (step - 3))
        pt.grad = g.clone()
        opt.step()
        gdev = g.to(DEV)
        L.check(L.lib().vidc_adam_step(L.ptr(p), L.ptr(gdev), L.ptr(mm), L.ptr(vv), 1000, 1e-2, 0.9, 0.999, 1e-8, step, L.current_stream()), "adam")
        g3 = rep(g).to(DEV)
        L.check(L.lib().vidc_adam_step(L.ptr(p3), L.ptr(g3), L.ptr(m3), L.ptr(v3), 1003, 1e-2, 0.9, 0.999, 1e-8, step, L.current_stream()), "adam")
        torch.cuda.synchronize()
    assert (p.cpu() - pt.detach()).abs().max() < 2e-6
    assert torch.equal(p3[:1000], p), "body"
    assert torch.equal(m3[1000:], mm[:3]) and torch.equal(v3[1000:], vv[:3]), "tail moments"
    assert torch.equal(p3[1000:], p[:3]), "tail parameters"


def _train_fixture(golden_dir):
    f = np.load(os.path.join(golden_dir, "train_step.npz"))
    batch = S.synthetic_batch(2, 240, 320, 1234, frame0=int(f["frame0"]))
    gt = S.synthetic_ground_truth_depth(batch["image"], 1234)
    din = torch.zeros(2, 240, 320)
    rc = torch.from_numpy(f["depth_in_rc"]).long()
    din[rc[:, 0], rc[:, 1], rc[:, 2]] = torch.from_numpy(f["depth_in_val"])
    return f, batch["image"], torch.from_numpy(f["normal"]), din[:, None], gt


@gpu
@pytest.mark.parametrize("k,stride,pad,C,H,W", [(3, 1, 1, 64, 9, 11), (3, 2, 1, 32, 10, 7), (1, 1, 0, 96, 5, 6), (1, 2, 0, 64, 7, 9), (3, 1, 1, 30, 6, 7), (1, 1, 0, 132, 9, 13)])
def test_im2col_transposed_operands(k, stride, pad, C, H, W):
    """The wgrad GEMM's operands (training.py `_wgrad_gemm`): xt[(tap*C + c)][m] is F.unfold's column matrix in (tap, channel) row order,
    rows padded with zeros to a multiple of 32 -- exact (pure data movement); with split=1 the same rows bit-identical to
    vidc_split_bf16x3 applied to the fp32 rows."""
    from vi_depth_completion_amd import _lib as L
    lib = L.lib()
    B = 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = B * Ho * Wo
    Mp = (M + 31) // 32 * 32
    cols = F.unfold(x, k, padding=pad, stride=stride)                       # (B, C*k*k, Ho*Wo), rows ordered (c, tap)
    want = cols.view(B, C, k * k, Ho * Wo).permute(2, 1, 0, 3).reshape(k * k * C, M)
    want = F.pad(want, (0, Mp - M))
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    xt = torch.full((k * k * C, Mp), float("nan"), device=DEV)
    st = L.current_stream()
    L.check(lib.vidc_im2col_transposed(L.ptr(xd), L.ptr(xt), B, H, W, C, C, Ho, Wo, k, k, stride, pad, Mp, 0, st), "im2col^T")
    assert torch.equal(xt.cpu(), want)
    xs = torch.empty_like(xt)
    L.check(lib.vidc_im2col_transposed(L.ptr(xd), L.ptr(xs), B, H, W, C, C, Ho, Wo, k, k, stride, pad, Mp, 1, st), "im2col^T split")
    ref = torch.empty_like(xt)
    L.check(lib.vidc_split_bf16x3(L.ptr(xt), L.ptr(ref), k * k * C, Mp, Mp, st), "split")
    assert torch.equal(xs.view(torch.int32).cpu(), ref.view(torch.int32).cpu())
    Mq = (M + 63) // 64 * 64                                               # plain bf16 rows (split = 2): Mp a multiple of 64
    xb = torch.empty(k * k * C, Mq // 2, device=DEV)
    L.check(lib.vidc_im2col_transposed(L.ptr(xd), L.ptr(xb), B, H, W, C, C, Ho, Wo, k, k, stride, pad, Mq, 2, st), "im2col^T bf16")
    assert torch.equal(xb.view(torch.bfloat16).cpu(), F.pad(want, (0, Mq - Mp)).to(torch.bfloat16))
    # split + 4: the same rows in channel-major order (c * taps + tap) = F.unfold's own row order: the wgrad GEMM then writes OIHW directly
    want_cm = F.pad(cols.permute(1, 0, 2).reshape(C * k * k, M), (0, Mp - M))
    for fmt, got_ref in ((0, xt), (1, xs)):
        xc = torch.full((k * k * C, Mp), float("nan"), device=DEV)
        L.check(lib.vidc_im2col_transposed(L.ptr(xd), L.ptr(xc), B, H, W, C, C, Ho, Wo, k, k, stride, pad, Mp, fmt | 4, st), "im2col^T channel-major")
        if fmt == 0:
            assert torch.equal(xc.cpu(), want_cm)
        perm = got_ref.view(k * k, C, Mp).permute(1, 0, 2).reshape(C * k * k, Mp)
        assert torch.equal(xc.view(torch.int32).cpu(), perm.contiguous().view(torch.int32).cpu())
    xcb = torch.empty(k * k * C, Mq // 2, device=DEV)
    L.check(lib.vidc_im2col_transposed(L.ptr(xd), L.ptr(xcb), B, H, W, C, C, Ho, Wo, k, k, stride, pad, Mq, 2 | 4, st), "im2col^T bf16 channel-major")
    assert torch.equal(xcb.view(torch.bfloat16).cpu(), F.pad(want_cm, (0, Mq - Mp)).to(torch.bfloat16))
    if C % 8 == 0:       # the same operand gathered from the dense bf16 copy of x (vidc_cast_bf16) instead of the fp32 tensor
        xbf = torch.empty(B, H, W, C // 2, device=DEV)
        L.check(lib.vidc_cast_bf16(L.ptr(xd), L.ptr(xbf), B * H * W, C, C, st), "cast")
        xcb2 = torch.full((k * k * C, Mq // 2), 9.0, device=DEV)
        L.check(lib.vidc_im2col_transposed_bf16(L.ptr(xbf), L.ptr(xcb2), B, H, W, C, Ho, Wo, k, k, stride, pad, Mq, st), "im2col^T from bf16")
        assert torch.equal(xcb2.view(torch.int32).cpu(), xcb.view(torch.int32).cpu())


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@gpu
@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,tile", [(64, 64, 3, 1, 1, 12, 20, 4), (128, 128, 1, 1, 0, 9, 7, 6), (192, 64, 3, 2, 1, 11, 13, 0), (64, 128, 3, 1, 1, 16, 16, 3),
                                                            # the pipelined-fragment-read tilings (SPEC 2): plain bf16 runs two MFMA passes per k-half there
                                                            (128, 128, 3, 1, 1, 17, 19, 33), (64, 192, 3, 1, 1, 16, 16, 34), (128, 64, 1, 1, 0, 9, 7, 35), (192, 128, 3, 2, 1, 21, 13, 36)])
def test_plain_bf16_conv_mode(cin, cout, k, stride, pad, H, W, tile):
    """VIDC_PREC_BF16 (the arithmetic BASELINE configs[4] names): operands rounded to bf16 by vidc_cast_bf16 / pack kinds 4, 5, products
    exact (bf16 x bf16 fits fp32), fp32 accumulation -- so the result equals F.conv2d of the bf16-ROUNDED operands in fp32 up to
    summation order (1e-5 of scale), forward weights and dgrad weights alike."""
    from vi_depth_completion_amd import _lib as L
    import ctypes as C
    lib, st = L.lib(), L.current_stream()
    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    want = F.conv2d(_bf16_round(x), _bf16_round(w), None, stride, pad)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    xb = torch.empty(B, H, W, cin // 2, device=DEV)
    L.check(lib.vidc_cast_bf16(L.ptr(xd), L.ptr(xb), B * H * W, cin, cin, st), "cast")
    assert torch.equal(xb.view(torch.bfloat16).cpu(), x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))
    wd = w.to(DEV)
    wp = torch.empty(w.numel() // 2, device=DEV)
    item = (L.PackItem * 1)()
    item[0].w, item[0].packed, item[0].Cout, item[0].Cin, item[0].KH, item[0].KW, item[0].kind, item[0].block_begin = L.ptr(wd), L.ptr(wp), cout, cin, k, k, 4, 0
    dev = torch.frombuffer(bytearray(bytes(item)), dtype=torch.uint8).to(DEV)
    L.check(lib.vidc_pack_conv_weights_batched(L.ptr(dev), 1, lib.vidc_pack_item_blocks(cout, cin, k, k, 4), st), "pack")
    y = torch.empty(B, Ho, Wo, cout, device=DEV)
    ones, zeros = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    d = L.ConvDesc()
    d.x, d.w, d.y, d.scale1, d.shift1 = L.ptr(xb), L.ptr(wp), L.ptr(y), L.ptr(ones), L.ptr(zeros)
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, cin // 2, cin // 2
    d.Ho, d.Wo, d.Cout, d.ldy = Ho, Wo, cout, cout
    d.KH, d.KW, d.stride, d.pad, d.flags = k, k, stride, pad, 0
    d.groups, d.splitk, d.precision, d.tile = 1, 1, L.PREC_BF16, tile
    d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin // 2, cout * k * k * cin // 2, cout, cout
    if tile == 0:
        L.check(lib.vidc_conv2d_plan(C.byref(d)), "plan")
        d.splitk = 1
    L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "conv bf16")
    _close(_nchw(y), want, 1e-5, "bf16 conv")
    # dgrad weights (kind 5): dx = conv_transpose of dy, stride 1 only here
    if stride == 1:
        gy = torch.randn(B, cout, Ho, Wo, generator=g)
        want_dx = F.conv_transpose2d(_bf16_round(gy), _bf16_round(w), None, 1, pad)
        item[0].kind = 5
        wq = torch.empty(w.numel() // 2, device=DEV)
        item[0].packed = L.ptr(wq)
        dev = torch.frombuffer(bytearray(bytes(item)), dtype=torch.uint8).to(DEV)
        L.check(lib.vidc_pack_conv_weights_batched(L.ptr(dev), 1, lib.vidc_pack_item_blocks(cout, cin, k, k, 5), st), "pack dgrad")
        gd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
        gb = torch.empty(B, Ho, Wo, cout // 2, device=DEV)
        L.check(lib.vidc_cast_bf16(L.ptr(gd), L.ptr(gb), B * Ho * Wo, cout, cout, st), "cast")
        dx = torch.empty(B, H, W, cin, device=DEV)
        o2, z2 = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
        e = L.ConvDesc()
        e.x, e.w, e.y, e.scale1, e.shift1 = L.ptr(gb), L.ptr(wq), L.ptr(dx), L.ptr(o2), L.ptr(z2)
        e.B, e.H, e.W, e.Cin, e.ldx = B, Ho, Wo, cout // 2, cout // 2
        e.Ho, e.Wo, e.Cout, e.ldy = H, W, cin, cin
        e.KH, e.KW, e.stride, e.pad, e.flags = k, k, 1, k - 1 - pad, 0
        e.groups, e.splitk, e.precision, e.tile = 1, 1, L.PREC_BF16, 4
        e.x_gs, e.w_gs, e.y_gs, e.p_gs = cout // 2, cin * k * k * cout // 2, cin, cin
        L.check(lib.vidc_conv2d_bn_act(C.byref(e), st), "dgrad bf16")
        _close(_nchw(dx), want_dx, 1e-5, "bf16 dgrad")


@gpu
def test_batched_weight_packing_equals_the_per_layer_packs():
    """vidc_pack_conv_weights_batched (one launch for the whole network, every step) against the per-layer entry points it replaces:
    forward / dgrad, fp32 / split-bf16 -- byte-identical."""
    from vi_depth_completion_amd import _lib as L
    lib, st = L.lib(), L.current_stream()
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 32, 3, 3), (96, 64, 1, 1), (32, 128, 3, 3), (256, 64, 1, 1), (64, 64, 3, 3)]
    ws = [torch.randn(*sh, generator=g).to(DEV) for sh in shapes]
    items, want = [], []
    for w in ws:
        co, ci, kh, kw = w.shape
        n = w.numel()
        for kind in range(4):
            ref = torch.empty(n, device=DEV)
            if kind == 0:
                L.check(lib.vidc_pack_conv_weight(L.ptr(w), L.ptr(ref), co, ci, kh, kw, st), "pack")
            elif kind == 2:
                L.check(lib.vidc_pack_conv_weight_bf16x3(L.ptr(w), L.ptr(ref), co, ci, kh, kw, st), "pack")
            else:
                tmp = torch.empty(n, device=DEV)
                L.check(lib.vidc_pack_conv_weight_dgrad(L.ptr(w), L.ptr(tmp if kind == 3 else ref), co, ci, kh, kw, st), "pack")
                if kind == 3:
                    L.check(lib.vidc_split_bf16x3(L.ptr(tmp), L.ptr(ref), ci, kh * kw * co, kh * kw * co, st), "split")
            want.append(ref)
            items.append((w, torch.full((n,), float("nan"), device=DEV), co, ci, kh, kw, kind))
    table = (L.PackItem * len(items))()
    blocks = 0
    for t, (w, out, co, ci, kh, kw, kind) in zip(table, items):
        t.w, t.packed, t.Cout, t.Cin, t.KH, t.KW, t.kind, t.block_begin = L.ptr(w), L.ptr(out), co, ci, kh, kw, kind, blocks
        blocks += lib.vidc_pack_item_blocks(co, ci, kh, kw, kind)
    dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
    L.check(lib.vidc_pack_conv_weights_batched(L.ptr(dev), len(items), blocks, st), "batched pack")
    for (w, out, *_rest), ref in zip(items, want):
        assert torch.equal(out.view(torch.int32).cpu(), ref.view(torch.int32).cpu()), "kind %d of %s" % (_rest[-1], tuple(w.shape))
    assert lib.vidc_pack_conv_weights_batched(None, 1, 1, st) != 0 and lib.vidc_pack_conv_weights_batched(L.ptr(dev), 0, 0, st) != 0


@gpu
@pytest.mark.parametrize("offset", [0, 1, 3])
def test_batched_pack_plain_bf16_layouts(offset):
    """Kinds 4 / 5 of the batched re-packing (plain bf16, the training mode of configs[4]) against the layouts include/vidc.h states,
    written out with torch: forward [co][ci/64][kh][kw][64] and dgrad [ci][co/64][KH-1-kh][KW-1-kw][64], round-to-nearest-even.  The
    parameters sit at 4-, 8- and 16-byte alignments (views of one flat buffer, as in the trainer), several items per launch."""
    from vi_depth_completion_amd import _lib as L
    lib, st = L.lib(), L.current_stream()
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 64, 1), (128, 64, 3), (64, 192, 1), (192, 128, 3), (256, 1024, 1), (1024, 256, 1), (64, 8, 3), (128, 72, 1), (64, 24, 3)]
    flat = torch.randn(offset + sum(co * ci * k * k + 5 for co, ci, k in shapes), generator=g).to(DEV)
    items, want, o = [], [], offset
    for co, ci, k in shapes:
        w = flat[o:o + co * ci * k * k].view(co, ci, k, k)
        o += w.numel() + 5
        for kind in (4, 5):
            if (ci if kind == 4 else co) % 64:
                assert lib.vidc_pack_item_blocks(co, ci, k, k, kind) == 0
                continue
            if kind == 4:
                ref = w.view(co, ci // 64, 64, k, k).permute(0, 1, 3, 4, 2)
            else:
                ref = w.flip(2, 3).reshape(co // 64, 64, ci, k, k).permute(2, 0, 3, 4, 1)
            want.append(ref.contiguous().to(torch.bfloat16).reshape(-1))
            items.append((w, co, ci, k, kind))
    outs = [torch.full((w.numel() // 2,), float("nan"), device=DEV) for w, *_ in items]
    table = (L.PackItem * len(items))()
    blocks = 0
    for t, (w, co, ci, k, kind), out in zip(table, items, outs):
        t.w, t.packed, t.Cout, t.Cin, t.KH, t.KW, t.kind, t.block_begin = L.ptr(w), L.ptr(out), co, ci, k, k, kind, blocks
        blocks += lib.vidc_pack_item_blocks(co, ci, k, k, kind)
    dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
    L.check(lib.vidc_pack_conv_weights_batched(L.ptr(dev), len(items), blocks, st), "batched pack")
    for (w, co, ci, k, kind), out, ref in zip(items, outs, want):
        got = out.view(torch.bfloat16)
        assert torch.equal(got.view(torch.int16).cpu(), ref.view(torch.int16).cpu()), "kind %d of %s" % (kind, tuple(w.shape))


@gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_training_iteration_vs_reference(golden_dir, seeded_weights, monkeypatch, precision):
    """(precision: the arithmetic of the forward / dgrad convs.  fp32 = exact products like the reference, the trainer's default: the
    tolerances below.  bf16x3 = VIDC_TRAIN_PRECISION=bf16x3, the split-bf16 3-pass mode of the inference path, ~2^-16 per product, a
    throughput option: same loss, same prediction, same global gradient norm, but single gradient tensors are only asserted to 1e-1
    of their scale (5e-3 in the head): a forward that differs by 1e-5 flips ReLU gates, and the flips propagate through 335
    train-mode BatchNorms; observed up to 6 %.)
    ONE `_run_training_iteration` (network_run.py:231-254) of the whole 310 M-parameter network on the reference's own 2-frame
    batch: the loss the reference logged, the gradient of 29 parameters from the stems to the head, the global gradient norm (1e-3),
    the parameters after the Adam step, the updated running statistics.
    Gradient tolerance: 2e-2 of each tensor's scale.  That is the noise floor of the REFERENCE's fp32 arithmetic, not slack: its own
    gradients differ from an fp64 evaluation of the same step by up to 1.0e-2 of scale on the stems and early stages (337 convolutions
    and 335 train-mode BatchNorms away from the loss; oracle/tools/train_fp64_check.py), by 1e-7..1e-3 in the decoder and head, where
    this test is as tight (5e-4).  Conv biases in front of a BatchNorm have gradient 0 in exact arithmetic: both sides hold noise."""
    from _probe import check_probe
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    monkeypatch.setenv("VIDC_TRAIN_PRECISION", precision)
    f, image, normal, depth_in, gt = _train_fixture(golden_dir)
    cnn = ModifiedFPN().to(DEV)
    st = cnn.state_dict()
    st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
    cnn.load_state_dict(st)
    cnn.train()
    tr = DepthCompletionTrainer(cnn, float(f["lr"]))
    loss, pred = tr.forward_backward(image.to(DEV), normal.to(DEV), depth_in.to(DEV), gt.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(f["loss"])) < 2e-5 * float(f["loss"]), (float(loss), float(f["loss"]))
    assert round(float(loss), 4) == float(f["loss_logged"])
    assert np.abs(pred[:, 0, ::16, ::16].cpu().numpy() - f["pred_probe"]).max() < 1e-3
    gn = float(torch.sqrt((tr.flat_g.double() ** 2).sum()))
    assert abs(gn - float(f["grad_global_norm"])) < 1e-3 * gn, (gn, float(f["grad_global_norm"]))
    names = sorted({k.split("|")[1] for k in f.files if k.startswith("grad|")})
    for k in names:
        if k in ("feature1_upsamping.0.bias", "feature4_upsamping.14.bias"):          # bias in front of a BatchNorm: exact gradient 0
            assert float(tr.grad[k].abs().max()) < 1e-5, k
            continue
        tight = k.startswith("feature_concat") or k in ("feature1_upsamping.4.weight", "feature1_upsamping.0.weight")
        check_probe(f, "grad", k, tr.grad[k].cpu(), (5e-4 if precision == "fp32" else 5e-3) if tight else (2e-2 if precision == "fp32" else 1e-1), 1e-7)
    tr.optimizer_step()
    torch.cuda.synchronize()
    # First Adam step: every weight moves by lr * g / (|g| + 1e-8) ~ lr * sign(g).  Where the gradient is well above its noise floor
    # (> 5 % of the tensor's largest entry) the sign is certain and the stepped value must match to 1e-6; elsewhere a sign flip of a
    # near-zero gradient moves the value by up to 2 * lr = 2e-4 on either side, which is all that can be asserted there.
    lr = float(f["lr"])
    n_certain = 0
    for k in names:
        key, t = "|%s|" % k, tr.param[k].cpu().reshape(-1)
        if "new" + key + "full" in f.files:
            new, gref, got = f["new" + key + "full"], f["grad" + key + "full"], t.numpy()
        else:
            new, gref, got = f["new" + key + "val"], f["grad" + key + "val"], t[torch.from_numpy(f["new" + key + "idx"])].numpy()
        assert np.abs(got - new).max() < 2.1 * lr, (k, np.abs(got - new).max())
        if np.abs(gref).max() < 1e-7:          # bias in front of a BatchNorm: the whole gradient is rounding noise around 0
            continue
        certain = np.abs(gref) > (0.05 if precision == "fp32" else 0.3) * np.abs(gref).max()
        n_certain += int(certain.sum())
        assert np.abs(got - new)[certain].max() < 1e-6, (k, np.abs(got - new)[certain].max())
        old = f["old" + key + ("full" if "old" + key + "full" in f.files else "val")]
        assert np.abs(np.abs(got - old)[certain] - lr).max() < 2e-6, k                    # ... and it did move by lr
    assert n_certain > (1000 if precision == "fp32" else 200)
    sd = cnn.state_dict()
    for k in [k[4:] for k in f.files if k.startswith("buf|")]:
        assert np.abs(sd[k].cpu().numpy() - f["buf|" + k]).max() < 1e-4 * max(1.0, np.abs(f["buf|" + k]).max()), k
    assert int(sd["resnet_rgb.bn1.num_batches_tracked"]) == 1
    # the module is usable for inference again after the step (derived weights invalidated)
    cnn.eval()
    out = cnn(image[:1].to(DEV), normal[:1].to(DEV), depth_in[:1].to(DEV))
    assert torch.isfinite(out).all()


@gpu
def test_plain_bf16_training_mode(golden_dir, seeded_weights, monkeypatch):
    """VIDC_TRAIN_PRECISION=bf16 -- what BASELINE configs[4] names ("bf16 ... MFMA fwd+bwd convs"; SURVEY §8 f3: "bf16 with fp32
    master"): conv operands rounded to bf16 (8-bit mantissa), fp32 accumulation, everything else fp32.  The reference has no such mode,
    so the bars are those of mixed-precision training against its fp32 iteration on the fixture batch, stated here: loss within 5e-3
    relative (observed 8e-4), global gradient norm within 2e-2 (observed 3e-4), the flat gradient's cosine with the fp32 mode's above
    0.99 (observed 0.998; per-block relative L2 up to 0.4 in the pyramids: bf16 noise flips ReLU gates under 335 train-mode
    BatchNorms -- tools/train_bf16_check.py prints the table), head gradients within 2e-2 of scale, and six steps on the batch bring
    the loss down like the fp32 mode's."""
    from _probe import check_probe
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    f, image, normal, depth_in, gt = _train_fixture(golden_dir)
    ins = [t.to(DEV) for t in (image, normal, depth_in, gt)]
    grads, losses = {}, {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("VIDC_TRAIN_PRECISION", mode)
        cnn = ModifiedFPN().to(DEV)
        st = cnn.state_dict()
        st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
        cnn.load_state_dict(st)
        cnn.train()
        tr = DepthCompletionTrainer(cnn, float(f["lr"]))
        loss, pred = tr.forward_backward(*ins)
        grads[mode] = tr.flat_g.double().clone()
        if mode == "bf16":
            assert abs(float(loss) - float(f["loss"])) < 5e-3 * float(f["loss"]), (float(loss), float(f["loss"]))
            gn = float(grads[mode].norm())
            assert abs(gn - float(f["grad_global_norm"])) < 2e-2 * gn
            for k in ("feature_concat.0.weight", "feature_concat.2.weight"):
                check_probe(f, "grad", k, tr.grad[k].cpu(), 2e-2, 1e-7)
        tr.optimizer_step()
        losses[mode] = [float(loss)] + [float(tr.step(*ins)) for _ in range(5)]
        del tr, cnn
        torch.cuda.empty_cache()
    cos = float((grads["fp32"] * grads["bf16"]).sum() / (grads["fp32"].norm() * grads["bf16"].norm()))
    print("bf16 vs fp32: gradient cosine %.5f; losses fp32 %s bf16 %s" % (cos, losses["fp32"], losses["bf16"]))
    assert cos > 0.99
    assert losses["bf16"][-1] < 0.9 * losses["bf16"][0]
    assert abs(losses["bf16"][-1] - losses["fp32"][-1]) < 0.05 * losses["fp32"][-1]


@gpu
def test_fused_transposed_gradient_of_the_bn_backward_is_bit_identical(golden_dir, seeded_weights, monkeypatch):
    """bf16 mode: the BatchNorm backward behind a conv also writes that conv's dY transposed as bf16 rows (vidc_bn_train_backward_t), the
    left operand of its weight-gradient GEMM, instead of a transpose launch per conv.  Same values rounded once from the same fp32
    result: the whole flat gradient and the loss are identical to the step with the separate transposes (VIDC_TRAIN_DYT_FUSED=0).
    The same comparison covers the weight-gradient GEMM writing the parameter's .grad in place (channel-major operand rows: its output is
    OIHW) against the staged form with a permute / copy launch per conv (VIDC_TRAIN_WGRAD_INPLACE=0).
    And the kernel alone against vidc_bn_train_backward + vidc_im2col_transposed on a ragged shape (M not a multiple of 64)."""
    import ctypes as C
    from vi_depth_completion_amd import _lib as L
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    lib = L.lib()
    B, H, W, Cc = 2, 9, 13, 192
    M, Mp = B * H * W, (B * H * W + 63) // 64 * 64
    g = torch.Generator().manual_seed(3)
    dy, x = torch.randn(B, H, W, Cc, generator=g).to(DEV), torch.randn(B, H, W, Cc, generator=g).to(DEV)
    yr = torch.randn(B, H, W, Cc, generator=g).to(DEV)
    gamma, mean, rstd = (torch.rand(Cc, generator=g) + 0.5).to(DEV), torch.randn(Cc, generator=g).to(DEV) * 0.1, (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    sc = torch.empty(lib.vidc_train_scratch_bytes(M, Cc), dtype=torch.uint8, device=DEV)
    outs = []
    for fused in (False, True):
        dx, dxb = torch.empty_like(dy), torch.zeros(B, H, W, Cc // 2, device=DEV)
        dxt = torch.full((Cc, Mp // 2), 7.0, device=DEV)
        dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
        L.check(lib.vidc_bn_train_backward_t(L.ptr(dy), L.ptr(x), L.ptr(yr), L.ptr(dx), M, Cc, Cc, Cc, Cc, Cc, L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                                             L.ptr(dg), L.ptr(db), L.ptr(dxb), L.ptr(dxt) if fused else None, Mp, L.ptr(sc), L.current_stream()), "bn_bwd_t")
        if not fused:
            L.check(lib.vidc_im2col_transposed(L.ptr(dx), L.ptr(dxt), B, H, W, Cc, Cc, H, W, 1, 1, 1, 0, Mp, 2, L.current_stream()), "transpose")
        outs.append([t.cpu() for t in (dx, dxb, dxt, dg, db)])
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # vidc_transpose_bf16 of the dense bf16 copy == vidc_im2col_transposed(split = 2) of the fp32 tensor it was rounded from
    xb = torch.empty(B, H, W, Cc // 2, device=DEV)
    L.check(lib.vidc_cast_bf16(L.ptr(x), L.ptr(xb), M, Cc, Cc, L.current_stream()), "cast")
    t_a, t_b = torch.full((Cc, Mp // 2), 3.0, device=DEV), torch.full((Cc, Mp // 2), 5.0, device=DEV)
    L.check(lib.vidc_transpose_bf16(L.ptr(xb), L.ptr(t_a), M, Cc, Mp, L.current_stream()), "transpose_bf16")
    L.check(lib.vidc_im2col_transposed(L.ptr(x), L.ptr(t_b), B, H, W, Cc, Cc, H, W, 1, 1, 1, 0, Mp, 2 | 4, L.current_stream()), "im2col^T")
    assert torch.equal(t_a.view(torch.int32).cpu(), t_b.view(torch.int32).cpu())
    f, image, normal, depth_in, gt = _train_fixture(golden_dir)
    ins = [t.to(DEV) for t in (image, normal, depth_in, gt)]
    monkeypatch.setenv("VIDC_TRAIN_PRECISION", "bf16")
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("VIDC_TRAIN_DYT_FUSED", fused)
        monkeypatch.setenv("VIDC_TRAIN_GROUPED", "0")              # (per-pyramid chains on both sides: the staged weight gradient has no grouped form)
        monkeypatch.setenv("VIDC_TRAIN_WGRAD_INPLACE", fused)      # (0: tap-major operand rows, staging buffer, permute / copy launches)
        monkeypatch.setenv("VIDC_TRAIN_XT_BF16", fused)            # (0: the 1x1 convs' right operand transposed from the fp32 tensor)
        monkeypatch.setenv("VIDC_TRAIN_SKIP_F32_DY", fused)        # (0: the BatchNorm backward also writes the fp32 dY nobody reads)
        monkeypatch.setenv("VIDC_TRAIN_BN_ADD_FUSED", fused)       # (0: relu(bn3(.) + identity) as a BatchNorm followed by an add kernel)
        monkeypatch.setenv("VIDC_TRAIN_ADD_BF16", fused)           # (0: the next block's convs cast the block output themselves)
        cnn = ModifiedFPN().to(DEV)
        st = cnn.state_dict()
        st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
        cnn.load_state_dict(st)
        cnn.train()
        tr = DepthCompletionTrainer(cnn, float(f["lr"]))
        loss, _ = tr.forward_backward(*ins)
        res[fused] = (float(loss), tr.flat_g.clone().cpu())
        del tr, cnn
        torch.cuda.empty_cache()
    assert res["1"][0] == res["0"][0]
    assert torch.equal(res["1"][1], res["0"][1])


@gpu
def test_training_loop_on_pipeline_inputs(seeded_weights):
    """The binding INTEGRATION.md shows: the pipeline produces what `_call_cnn` feeds the depth network (image, predicted normals,
    enriched depth), the trainer runs `_run_training_iteration` on it; four iterations on one batch bring the loss down, the parameters
    stay views of the flat buffer, and the stepped network serves inference again (BatchNorm running statistics included)."""
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    pipe = DepthCompletionPipeline(enriched_samples=200, rng=np.random.RandomState(3))
    pipe.load_state_dicts(seeded_weights["sn"], seeded_weights["dc"])
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    batch = S.synthetic_batch(2, 240, 320, 1234, frame0=50)
    gt = S.synthetic_ground_truth_depth(batch["image"], 1234).to(DEV)
    rgb, normals, depth_in = pipe.network_inputs(batch)
    assert rgb.shape == (2, 3, 240, 320) and normals.shape == (2, 3, 240, 320) and depth_in.shape == (2, 1, 240, 320)
    assert int((depth_in > 0).sum()) > int((batch["sparse_depth"] > 0).sum())          # enriched
    before = pipe._call_cnn(batch).clone()
    pipe.cnn.train()
    tr = DepthCompletionTrainer(pipe.cnn, 1e-4)
    losses = [float(tr.step(rgb, normals, depth_in, gt)) for _ in range(4)]
    print("losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    p = dict(pipe.cnn.named_parameters())["feature_concat.0.weight"]
    assert p.data_ptr() == tr.param["feature_concat.0.weight"].data_ptr() and p.grad.data_ptr() == tr.grad["feature_concat.0.weight"].data_ptr()
    pipe.cnn.eval()
    pipe.rng = np.random.RandomState(3)
    after = pipe._call_cnn(batch)
    assert torch.isfinite(after).all() and float((after - before).abs().mean()) > 1e-4      # the trained weights are the ones that run
    # the layout travels with the optimizer state, and a network whose tensors were rebound after the trainer was built is refused at the next step
    # instead of being read through stale base pointers
    st = tr.flat_state()
    assert st["layout"] == tr.layout == "per_pyramid" and st["order"][0] == tr.named[0][0]
    tr.load_flat_state(st)
    with pytest.raises(RuntimeError, match="layout"):
        tr.load_flat_state(dict(st, layout="grouped"))
    pipe.cnn.train()
    first = tr.named[0][1]
    first.data = first.data.clone()            # what cnn.float() / .to() do: a new storage behind the same Parameter
    with pytest.raises(RuntimeError, match="rebound"):
        tr.step(rgb, normals, depth_in, gt)


@gpu
def test_graph_replay_equals_eager_steps(seeded_weights):
    """`step()` replays forward + backward as one captured hipGraph from the third step of a shape on: five steps on changing inputs,
    graph on against graph off, and stream lanes on against one stream -- the same kernels with fixed-order reductions, so losses,
    parameters and running statistics are bit-identical."""
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    runs = []
    for use_graph, lanes in ((False, 4), (True, 4), (False, 1)):      # (the last run: everything on ONE stream)
        cnn = ModifiedFPN().to(DEV)
        cnn.load_state_dict({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
        cnn.train()
        tr = DepthCompletionTrainer(cnn, 1e-4)
        tr.use_graph, tr.n_lanes = use_graph, lanes
        losses = []
        for it in range(5):
            b = S.synthetic_batch(1, 96, 128, 77, frame0=it)
            image = b["image"].to(DEV)
            normal = F.normalize(image - 0.5, dim=1)
            gt = S.synthetic_ground_truth_depth(b["image"], 77).to(DEV)
            losses.append(float(tr.step(image, normal, b["sparse_depth"].to(DEV), gt)))
        assert bool(tr._graphs) == use_graph
        runs.append((losses, tr.flat_p.clone(), {k: v.clone() for k, v in cnn.state_dict().items() if "running" in k or "tracked" in k}))
    (l0, p0, s0), (l1, p1, s1), (l2, p2, s2) = runs
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)
    assert all(torch.equal(s0[k], s1[k]) for k in s0)
    # the three pyramids / four decoder branches on their own HIP streams against everything on one stream: the same kernels, per-lane
    # scratch and split-K workspaces -- bit-identical as well
    assert l0 == l2, (l0, l2)
    assert torch.equal(p0, p2) and all(torch.equal(s0[k], s2[k]) for k in s0)
    assert int(s1["resnet_rgb.bn1.num_batches_tracked"]) == 5 if "resnet_rgb.bn1.num_batches_tracked" in s1 else True


@gpu
def test_two_training_iterations_vs_oracle(golden_dir, seeded_weights):
    """Two consecutive iterations against oracle/train_oracle.py (itself pinned by the reference's first iteration): the second loss
    depends on the first Adam step, the second step on the moment estimates and bias corrections (step = 2), and the second forward on
    nothing stale (packed weights are rebuilt every step)."""
    from oracle import train_oracle as T
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    f, image, normal, depth_in, gt = _train_fixture(golden_dir)
    image, normal, depth_in, gt = image[:1], normal[:1], depth_in[:1], gt[:1]             # one frame keeps the CPU side short
    lr = 1e-3
    sd, state, ref_losses = dict(seeded_weights["dc"]), {}, []
    for _ in range(2):
        loss, sd = T.training_iteration(sd, image, normal, depth_in, gt, lr, state)
        ref_losses.append(float(loss))
    cnn = ModifiedFPN().to(DEV)
    st = cnn.state_dict()
    st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
    cnn.load_state_dict(st)
    cnn.train()
    tr = DepthCompletionTrainer(cnn, lr)
    got = [float(tr.step(image.to(DEV), normal.to(DEV), depth_in.to(DEV), gt.to(DEV))) for _ in range(2)]
    print("losses", got, ref_losses)
    assert abs(got[0] - ref_losses[0]) < 1e-5 * ref_losses[0]
    assert abs(got[1] - ref_losses[1]) < 2e-3 * ref_losses[1]          # one lr = 1e-3 step through 310 M weights, signs of ~0 gradients free
    assert ref_losses[1] != ref_losses[0]
    new = cnn.state_dict()
    for k in ("feature_concat.2.weight", "feature_concat.0.bias"):
        ref = sd[k].numpy()
        assert np.abs(new[k].cpu().numpy() - ref).max() < 2.1 * 2 * lr + 1e-4 * np.abs(ref).max(), k
    for k in ("resnet_rgb.conv1.bn_2.running_var", "feature4_upsamping.1.running_mean"):      # second-forward statistics: downstream of the
        ref = sd[k].numpy()                                                                  # +-lr sign noise of the first step
        assert np.abs(new[k].cpu().numpy() - ref).max() < 3e-2 * np.abs(ref).max(), k
    assert int(new["resnet_rgb.bn1.num_batches_tracked"]) == 2


# ---- two ranks through whole training steps --------------------------------------------------------------------------
def _run_two_rank_worker(extra_env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VIDC_TRAIN_PRECISION")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
                        os.path.join(root, "tests", "two_rank_training_worker.py")], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "TWO_RANK_TRAINING_OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
    return r.stdout


@gpu
def test_two_ranks_through_training_steps():
    """Two ranks (gloo, sharing the GPU), different shards, four `DepthCompletionTrainer.step`s each -- two eager, two as captured graphs,
    the decoder's gradient buckets all-reduced before the pyramids' backward has run: the all-reduced flat gradient equals the sum of
    the two shards' single-process gradients bit for bit, and each rank's parameters and per-rank BatchNorm statistics equal those of a
    single-process replica stepped with that sum (tests/two_rank_training_worker.py; replaces network_run.py:97-99)."""
    out = _run_two_rank_worker({})
    assert "bf16_buckets=0" in out


@gpu
def test_two_ranks_with_bf16_gradient_buckets():
    """The same with VIDC_TRAIN_GRAD_BF16=1: every bucket is narrowed to bf16 on the device, summed, widened back -- exactly
    round_bf16(round_bf16(g0) + round_bf16(g1)) in every element, half the bytes on the wire."""
    out = _run_two_rank_worker({"VIDC_TRAIN_GRAD_BF16": "1"})
    assert "bf16_buckets=1" in out


@gpu
def test_bf16_gradient_buckets_keep_the_loss_curve(seeded_weights):
    """What rounding the gradients to bf16 before Adam does to training (a world of one sends them through the narrow / widen kernels and
    the process group just the same): eight steps of the configs[4] workload at batch 2, bf16 convs, with and without -- the losses
    stay within 5e-3 relative of each other over the eight steps (measured 1.5e-3 at step 5, 4e-5 at step 2: Adam normalises the step by
    the gradient's own running magnitude, so 8 bits of mantissa per element do not move the direction; what grows from step to step is
    the bf16 conv arithmetic amplifying the first difference, as it does between two fp32 / bf16 runs of the same steps)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, socket
sys.path.insert(0, %r)
import torch
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", VIDC_DIST_WORLD1="1")
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from vi_depth_completion_amd import synthetic as S
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
from vi_depth_completion_amd.training import DepthCompletionTrainer
b = S.synthetic_batch(2, 240, 320, 1234, frame0=0)
image = b["image"].to(dev); normal = torch.nn.functional.normalize(image - 0.5, dim=1)
depth_in = b["sparse_depth"].to(dev); gt = S.synthetic_ground_truth_depth(b["image"], 1234).to(dev)
curves = []
for flag in ("0", "1"):
    os.environ["VIDC_TRAIN_GRAD_BF16"] = flag
    cnn = ModifiedFPN().to(dev)
    cnn.load_state_dict(S.seeded_state_dict(cnn.state_dict(), 1234, device=dev))
    cnn.train()
    tr = DepthCompletionTrainer(cnn, 1e-4)
    assert tr._distributed() and (tr.buckets.compress == "bf16") == (flag == "1")
    curves.append([float(tr.step(image, normal, depth_in, gt)) for _ in range(8)])
    del tr, cnn
print("CURVES", curves)
rel = max(abs(a - b) / abs(a) for a, b in zip(*curves))
assert curves[0][0] == curves[1][0] and curves[0][-1] < curves[0][0], curves      # same first loss (same weights), and it trains
assert rel < 5e-3, (rel, curves)
assert curves[0] != curves[1], "the switch changed nothing"
dist.barrier(); dist.destroy_process_group()
print("BF16_BUCKETS_CURVE_OK", rel)
''' % root
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", VIDC_TRAIN_PRECISION="bf16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "BF16_BUCKETS_CURVE_OK" in r.stdout, (r.stdout + r.stderr)[-4000:]


@gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_grouped_pyramids_are_bit_identical_to_per_pyramid_launches(golden_dir, seeded_weights, monkeypatch, precision):
    """Round 5: the same layer of the three ResNet-101 pyramids runs as ONE launch with three groups (forward conv, data gradient,
    weight-gradient GEMM) and ONE BatchNorm launch over the 3 x C channels; parameters, gradients, Adam moments and running statistics
    of the three layers sit next to each other in the flat buffers.  A group only selects base pointers: with the same tile / split-K per
    launch (forced here through the trainer's tune_hook: the measured table holds different entries for 1 and 3 groups) the loss, every
    gradient tensor, the running statistics and the parameters after the Adam step are bit-identical to the per-pyramid launch chains
    (VIDC_TRAIN_GROUPED=0) -- and the grouped step issues far fewer conv / BatchNorm launches."""
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    f, image, normal, depth_in, gt = _train_fixture(golden_dir)
    ins = [t.to(DEV) for t in (image, normal, depth_in, gt)]
    monkeypatch.setenv("VIDC_TRAIN_PRECISION", precision)
    monkeypatch.setenv("VIDC_TRAIN_GRAPH", "0")
    runs, launches = {}, {}
    for grouped in ("1", "0"):
        monkeypatch.setenv("VIDC_TRAIN_GROUPED", grouped)
        cnn = ModifiedFPN().to(DEV)
        st = cnn.state_dict()
        st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
        cnn.load_state_dict(st)
        cnn.train()
        tr = DepthCompletionTrainer(cnn, float(f["lr"]))
        assert tr.grouped == (grouped == "1")
        count = [0]

        def hook(d, role, count=count):
            count[0] += 1
            d.tile, d.splitk = 4, 1              # 64x64, no split-K: the same summation order whatever the group count
            return True
        tr.tune_hook = hook
        losses = [float(tr.step(*ins)) for _ in range(2)]
        launches[grouped] = count[0]
        runs[grouped] = (losses, {k: v.clone().cpu() for k, v in tr.grad.items()}, {k: v.clone().cpu() for k, v in cnn.state_dict().items()})
        del tr, cnn
        torch.cuda.empty_cache()
    assert runs["1"][0] == runs["0"][0], (runs["1"][0], runs["0"][0])
    for k, v in runs["1"][1].items():
        assert torch.equal(v.view(torch.int32), runs["0"][1][k].view(torch.int32)), "gradient of " + k
    for k, v in runs["1"][2].items():
        assert torch.equal(v, runs["0"][2][k]), "state_dict entry " + k
    # conv-kernel launches of a step (forward + dgrad + wgrad GEMM): 3 x 104 pyramid layers collapse to 104
    assert launches["1"] < 0.5 * launches["0"], launches


@gpu
def test_multi_rank_step_does_not_depend_on_the_hardware_queue_count():
    """Across ranks the step is two captured graphs (cut at the decoder's all-reduce).  Round 4 found that form 2x slower whenever the
    runtime exposes more than 4 hardware queues (GPU_MAX_HW_QUEUES, a default nobody controls on a shared node).  The multi-rank default
    is therefore the grouped single-stream chain (VIDC_TRAIN_GROUPED=auto): the configs[4] bench line through RCCL on a world of one
    (the same two graphs) at 8 queues must stay within 15 % of the 4-queue run; the per-pyramid lanes are shown to fall off the cliff."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def ms(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VIDC_TRAIN_GROUPED", "VIDC_TRAIN_STREAMS")}
        e.update(VIDC_DIST_WORLD1="1", VIDC_TRAIN_PRECISION="bf16", HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--train", "--batch", "8", "--steps", "10", "--warmup", "3"], env=e,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert "through the backend" in line["config"]["collectives"]
        return line["ms_per_step"]

    q4, q8 = ms(GPU_MAX_HW_QUEUES="4"), ms(GPU_MAX_HW_QUEUES="8")
    lanes8 = ms(GPU_MAX_HW_QUEUES="8", VIDC_TRAIN_GROUPED="0")
    print("two-graph step, bf16, batch 8: default (grouped, one stream) %.2f ms at 4 queues, %.2f at 8; per-pyramid lanes at 8 queues %.2f" % (q4, q8, lanes8))
    assert q8 < 1.15 * q4, (q4, q8)
    assert q8 < 35.0, q8


@gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_folded_batchnorm_reduction_is_bit_identical(golden_dir, seeded_weights, monkeypatch, precision):
    """Round 4 (an opt-in: exact, but measured slower -- csrc/train.hip fold_bn): the final reduction of a BatchNorm's chunk sums can run
    in the prologue of the kernel that consumes it (the apply pass of the forward, the dx pass of the backward) instead of in a launch
    of its own, on the maps small enough for the redundant reads to be cheap (vidc_train_bn_fold).  Kernel level on ragged shapes on both sides of that bound, every output compared bit
    for bit -- y, the bf16 copy, saved mean / invstd, updated running statistics, dx in its three forms, dgamma, dbeta -- then ONE whole
    training step: same loss, same flat gradient, same running statistics, in the exact-fp32 mode and in bf16."""
    from vi_depth_completion_amd import _lib as L
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    lib = L.lib()
    assert lib.vidc_train_bn_fold(-1) == 0, "the separate launch is the default"
    try:
        for (B, H, W, Cc, relu, with_res) in ((2, 9, 13, 192, 1, True), (8, 15, 20, 256, 1, False), (1, 7, 5, 64, 0, False), (8, 60, 80, 64, 1, True)):
            M, Mp = B * H * W, (B * H * W + 63) // 64 * 64
            g = torch.Generator().manual_seed(11)
            x, dy = torch.randn(B, H, W, Cc, generator=g).to(DEV) * 2 + 0.3, torch.randn(B, H, W, Cc, generator=g).to(DEV)
            res = torch.randn(B, H, W, Cc, generator=g).to(DEV) if with_res else None
            gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(DEV), torch.randn(Cc, generator=g).to(DEV) * 0.1
            sc = torch.empty(lib.vidc_train_scratch_bytes(M, Cc), dtype=torch.uint8, device=DEV)
            outs = []
            for fold in (0, 1):
                lib.vidc_train_bn_fold(fold)
                y, yb = torch.empty_like(x), torch.zeros(B, H, W, Cc // 2, device=DEV)
                mean, rstd = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
                rm, rv = torch.full((Cc,), 0.25, device=DEV), torch.full((Cc,), 1.5, device=DEV)
                L.check(lib.vidc_bn_train_forward_add(L.ptr(x), L.ptr(y), M, Cc, Cc, Cc, L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), 1e-5, 0.1, relu,
                                                      L.ptr(mean), L.ptr(rstd), L.ptr(yb), L.ptr(res) if with_res else None, Cc, L.ptr(sc), L.current_stream()), "bn fwd")
                got = [y, yb, mean, rstd, rm, rv]
                for transposed in (False, True):
                    dx, dxb = torch.empty_like(dy), torch.zeros(B, H, W, Cc // 2, device=DEV)
                    dxt = torch.full((Cc, Mp // 2), 7.0, device=DEV)
                    dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
                    L.check(lib.vidc_bn_train_backward_t(L.ptr(dy), L.ptr(x), L.ptr(y) if relu else None, L.ptr(dx), M, Cc, Cc, Cc, Cc, Cc, L.ptr(gamma), L.ptr(mean),
                                                         L.ptr(rstd), L.ptr(dg), L.ptr(db), L.ptr(dxb), L.ptr(dxt) if transposed else None, Mp, L.ptr(sc),
                                                         L.current_stream()), "bn bwd")
                    got += [dx, dxb, dg, db] + ([dxt] if transposed else [])
                outs.append([t.cpu() for t in got])
            for i, (a, b) in enumerate(zip(*outs)):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), ((B, H, W, Cc), i)
            ref = torch.nn.functional.batch_norm(x.permute(0, 3, 1, 2).cpu(), None, None, gamma.cpu(), beta.cpu(), True, 0.1, 1e-5).permute(0, 2, 3, 1)
            if with_res:
                ref = ref + res.cpu()
            ref = ref.clamp_min(0) if relu else ref
            assert float((outs[1][0] - ref).abs().max()) < 2e-5
        f, image, normal, depth_in, gt = _train_fixture(golden_dir)
        ins = [t.to(DEV) for t in (image, normal, depth_in, gt)]
        monkeypatch.setenv("VIDC_TRAIN_PRECISION", precision)
        runs = {}
        for fold in (1, 0):
            lib.vidc_train_bn_fold(fold)
            cnn = ModifiedFPN().to(DEV)
            st = cnn.state_dict()
            st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
            cnn.load_state_dict(st)
            cnn.train()
            tr = DepthCompletionTrainer(cnn, float(f["lr"]))
            loss, _ = tr.forward_backward(*ins)
            runs[fold] = (float(loss), tr.flat_g.clone().cpu(), {k: v.clone().cpu() for k, v in cnn.state_dict().items() if "running" in k})
            del tr, cnn
            torch.cuda.empty_cache()
        assert runs[1][0] == runs[0][0]
        assert torch.equal(runs[1][1], runs[0][1])
        for k, v in runs[1][2].items():
            assert torch.equal(v, runs[0][2][k]), k
    finally:
        lib.vidc_train_bn_fold(0)


@gpu
@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,tile,splitk", [(64, 64, 3, 1, 1, 12, 20, 4, 1), (128, 128, 1, 1, 0, 9, 7, 6, 1), (192, 64, 3, 2, 1, 11, 13, 2, 1),
                                                                   (64, 128, 3, 1, 1, 16, 16, 1, 2), (128, 96, 3, 1, 1, 15, 20, 29, 3), (64, 256, 1, 1, 0, 15, 20, 28, 1),
                                                                   (128, 128, 3, 1, 1, 17, 19, 33, 1), (192, 128, 3, 2, 1, 21, 13, 36, 2), (64, 64, 3, 1, 1, 30, 40, 26, 1)])
def test_conv_epilogue_channel_sums(cin, cout, k, stride, pad, H, W, tile, splitk):
    """VIDC_STATS_OUT (round 4, plain-bf16 training mode): the conv's epilogue also writes, per block of 32 output rows, the per-channel
    sum and sum of squares of the fp32 values it stores -- the partials of the train-mode BatchNorm behind it.  Against fp64 sums of the
    conv's OWN output (bias included), block by block: equal up to fp64 rounding (the summation order inside a block differs), for M not
    a multiple of 32 / of the tile, split-K (only the workgroup that finishes a tile contributes), loader-wave and pipelined tilings.
    Then `vidc_bn_train_forward_stats` on those partials against `vidc_bn_train_forward_add` on the same tensor."""
    from vi_depth_completion_amd import _lib as L
    import ctypes as C
    lib, st = L.lib(), L.current_stream()
    g = torch.Generator().manual_seed(5)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    bias = torch.randn(cout, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = B * Ho * Wo
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    xb = torch.empty(B, H, W, cin // 2, device=DEV)
    L.check(lib.vidc_cast_bf16(L.ptr(xd), L.ptr(xb), B * H * W, cin, cin, st), "cast")
    wd = w.to(DEV)
    wp = torch.empty(w.numel() // 2, device=DEV)
    item = (L.PackItem * 1)()
    item[0].w, item[0].packed, item[0].Cout, item[0].Cin, item[0].KH, item[0].KW, item[0].kind, item[0].block_begin = L.ptr(wd), L.ptr(wp), cout, cin, k, k, 4, 0
    dev = torch.frombuffer(bytearray(bytes(item)), dtype=torch.uint8).to(DEV)
    L.check(lib.vidc_pack_conv_weights_batched(L.ptr(dev), 1, lib.vidc_pack_item_blocks(cout, cin, k, k, 4), st), "pack")
    ones = torch.ones(cout, device=DEV)
    nch = (M + 31) // 32
    outs = []
    for with_stats in (False, True):
        y = torch.empty(B, Ho, Wo, cout, device=DEV)
        stats = torch.full((nch, 2, cout), float("nan"), dtype=torch.float64, device=DEV)
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale1, d.shift1 = L.ptr(xb), L.ptr(wp), L.ptr(y), L.ptr(ones), L.ptr(bias)
        d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, cin // 2, cin // 2
        d.Ho, d.Wo, d.Cout, d.ldy = Ho, Wo, cout, cout
        d.KH, d.KW, d.stride, d.pad = k, k, stride, pad
        d.flags = L.STATS_OUT if with_stats else 0
        d.y_split = L.ptr(stats) if with_stats else None
        d.groups, d.splitk, d.precision, d.tile = 1, splitk, L.PREC_BF16, tile
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin // 2, cout * k * k * cin // 2, cout, cout
        if splitk > 1:
            ws = torch.zeros(lib.vidc_conv2d_workspace_bytes(C.byref(d)) // 4 + 4, device=DEV)
            d.workspace = L.ptr(ws)
        L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "conv bf16 + stats")
        outs.append((y.cpu(), stats.cpu()))
    (y0, _), (y1, stats) = outs
    assert torch.equal(y0, y1), "the flag must not change the conv's output"
    assert torch.isfinite(stats).all(), "every block of every channel is written exactly by the workgroup that finishes its tile"
    rows = y1.reshape(M, cout).double()
    pad_rows = torch.cat([rows, torch.zeros(nch * 32 - M, cout, dtype=torch.float64)]).reshape(nch, 32, cout)
    want = torch.stack([pad_rows.sum(1), (pad_rows * pad_rows).sum(1)], dim=1)
    scale = want.abs().max().item()
    assert float((stats - want).abs().max()) < 1e-12 * max(scale, 1.0), float((stats - want).abs().max())
    # the BatchNorm forward on the conv's partials against the one that makes its own
    yd = y1.to(DEV)
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV) * 0.1
    res = []
    for use_stats in (False, True):
        z = torch.empty_like(yd)
        mean, rstd = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
        rm, rv = torch.full((cout,), 0.25, device=DEV), torch.full((cout,), 1.5, device=DEV)
        sc = torch.empty(lib.vidc_train_scratch_bytes(M, cout), dtype=torch.uint8, device=DEV)
        if use_stats:
            L.check(lib.vidc_bn_train_forward_stats(L.ptr(yd), L.ptr(z), M, cout, cout, cout, L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), 1e-5, 0.1, 1,
                                                    L.ptr(mean), L.ptr(rstd), None, None, 0, L.ptr(stats.to(DEV)), L.ptr(sc), st), "bn stats")
        else:
            L.check(lib.vidc_bn_train_forward_add(L.ptr(yd), L.ptr(z), M, cout, cout, cout, L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), 1e-5, 0.1, 1,
                                                  L.ptr(mean), L.ptr(rstd), None, None, 0, L.ptr(sc), st), "bn")
        res.append([t.cpu() for t in (z, mean, rstd, rm, rv)])
    for name, a, b in zip(("y", "mean", "rstd", "running_mean", "running_var"), *res):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (name, float((a - b).abs().max()))
    with pytest.raises(RuntimeError, match="STATS_OUT"):       # not offered outside the plain-bf16 mode
        d.precision = L.PREC_FP32
        L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "conv")


@gpu
def test_training_step_with_conv_epilogue_statistics(golden_dir, seeded_weights, monkeypatch):
    """One whole bf16 training step with the BatchNorm statistics taken from the convs' epilogues (the default) against the same step
    with every BatchNorm summing its input itself (VIDC_TRAIN_CONV_STATS=0).  The sums are the same numbers added in another order
    (fp64): loss equal to 1e-6 relative, flat gradient within 2e-3 of its scale in every element and 1e-5 in the mean (the bf16 convs
    amplify a last-bit difference of a mean / invstd; same magnitudes as between two legal fp64 orders), running statistics 1e-6."""
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    f, image, normal, depth_in, gt = _train_fixture(golden_dir)
    ins = [t.to(DEV) for t in (image, normal, depth_in, gt)]
    monkeypatch.setenv("VIDC_TRAIN_PRECISION", "bf16")
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VIDC_TRAIN_CONV_STATS", flag)
        cnn = ModifiedFPN().to(DEV)
        st = cnn.state_dict()
        st.update({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
        cnn.load_state_dict(st)
        cnn.train()
        tr = DepthCompletionTrainer(cnn, float(f["lr"]))
        assert tr.conv_stats == (flag == "1")
        loss, _ = tr.forward_backward(*ins)
        runs[flag] = (float(loss), tr.flat_g.clone().cpu(), {k: v.clone().cpu() for k, v in cnn.state_dict().items() if "running" in k})
        del tr, cnn
        torch.cuda.empty_cache()
    la, lb = runs["1"][0], runs["0"][0]
    assert abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    ga, gb = runs["1"][1], runs["0"][1]
    scale = float(gb.abs().max())
    assert float((ga - gb).abs().max()) < 2e-3 * scale and float((ga - gb).abs().mean()) < 1e-5 * scale, (float((ga - gb).abs().max()), float((ga - gb).abs().mean()), scale)
    for k, v in runs["1"][2].items():
        assert float((v - runs["0"][2][k]).abs().max()) <= 1e-6 * max(1.0, float(v.abs().max())), k
