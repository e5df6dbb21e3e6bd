"""SurfaceNormalDORN (networks/surface_normal_dorn.py; the --use_gravity 0 branch): oracle pinned to the reference's
output (golden made by importing the reference), parameter layout, and the HIP program against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import vidc_oracle as O
from vi_depth_completion_amd import synthetic as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dorn_synthetic.npz")
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def dorn_weights():
    f = np.load(GOLDEN)
    shapes = {k: torch.empty(eval(s), device="meta") for k, s in zip(f["keys"], f["shapes"])}
    return S.seeded_state_dict(shapes, 1234)


def test_dorn_oracle_reproduces_reference(dorn_weights):
    f = np.load(GOLDEN)
    x = S.synthetic_batch(1, 240, 320, 1234, frame0=5)["image"]
    taps = {}
    out = O.dorn_forward(dorn_weights, x, taps=taps)
    assert np.array_equal(taps["features"].reshape(-1)[torch.from_numpy(f["feat_idx"])].numpy(), f["feat_val"])
    assert np.array_equal(out[0, :, ::16, ::16].numpy(), f["normals_probe"])
    assert np.abs(out[0].numpy() - f["normals_f16"].astype(np.float32)).max() < 1e-3          # fp16 copy of the full map
    assert np.allclose(out.double().sum(dim=(0, 2, 3)).numpy(), f["normals_sum"], rtol=0, atol=1e-6)


def test_dorn_state_dict_keys_match_reference():
    from vi_depth_completion_amd.networks.surface_normal_dorn import SurfaceNormalDORN
    f = np.load(GOLDEN)
    sd = SurfaceNormalDORN().state_dict()
    assert list(sd.keys()) == list(f["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(f["shapes"])


def test_dorn_program_recording():
    """Dry-run on CPU: dilated ASPP convs, the Linear as a 1x1 conv over the flattened NHWC map, concat by channel slices."""
    from vi_depth_completion_amd.networks.surface_normal_dorn import SurfaceNormalDORN
    net = SurfaceNormalDORN().eval()
    prog = net.build_program(1, 240, 320, torch.device("cpu"), dry_run=True)
    kinds = [k for k, _, _, _ in prog.ops]
    assert kinds.count("avgpool") == 1 and kinds.count("normalize") == 1 and kinds.count("head") == 1 and kinds.count("stem") == 1
    dil = sorted(kw["dilation"] for k, _, _, kw in prog.ops if k == "conv" and kw.get("dilation", 1) > 1)
    assert dil == [6, 12, 18]
    lin = [kw for k, _, _, kw in prog.ops if k == "conv" and kw["keys"][0].startswith("aspp_module.encoder.global_fc@hwc")]
    assert len(lin) == 1 and lin[0]["geom"][:2] == (512, 2048 * 4 * 5) and lin[0]["x"].H == 1
    # the five branches write the five 512-channel slices of one 2560-channel buffer
    cat = prog.taps["concat"]
    offs = sorted(w_kw["y"].ch_off for k, _, w, w_kw in prog.ops if k in ("conv", "upsample") and w_kw["y"].buf == cat.buf)
    assert offs == [0, 512, 1024, 1536, 2048]
    # features at 1/8 resolution
    assert (prog.taps["features"].H, prog.taps["features"].W, prog.taps["features"].C) == (30, 40, 2048)


@gpu
def test_hip_dorn_vs_oracle(dorn_weights):
    """Unit normals: max |diff| 2e-3, mean 5e-5 (bf16x3 / fp32 mixed arithmetic over 107 sequential convs)."""
    from vi_depth_completion_amd.networks.surface_normal_dorn import SurfaceNormalDORN
    x = S.synthetic_batch(2, 240, 320, 1234, frame0=5)["image"]
    net = SurfaceNormalDORN().cuda().eval()
    net.load_state_dict(dorn_weights)
    with torch.no_grad():
        got = net(x.cuda()).cpu()
    want = O.dorn_forward(dorn_weights, x)
    d = (got - want).abs()
    assert got.shape == (2, 3, 240, 320)
    assert d.max() < 2e-3 and d.mean() < 5e-5, (float(d.max()), float(d.mean()))
    assert (got.norm(dim=1) - 1).abs().max() < 1e-5


@gpu
def test_hip_dilated_conv_and_avgpool():
    """The two new primitives against torch CPU: 3x3 conv with dilation 6 / 12 / 18 (padding = dilation) and AvgPool2d(8, 8, (1, 0))."""
    import ctypes as C
    import torch.nn.functional as F
    from vi_depth_completion_amd import _lib as L, ops
    x = S.normal01(3, "dil.x", (1, 64, 30, 40)).float()
    w = S.normal01(3, "dil.w", (64, 64, 3, 3), scale=0.05).float()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = ops.pack_conv_weight(w.cuda())
    s1, b1 = torch.ones(1, 64).cuda(), torch.zeros(1, 64).cuda()
    for dil in (6, 12, 18):
        ref = F.conv2d(x, w, None, 1, dil, dil)
        y = torch.empty(1, 30, 40, 64, device="cuda")
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale1, d.shift1 = xd.data_ptr(), wp.data_ptr(), y.data_ptr(), s1.data_ptr(), b1.data_ptr()
        d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 1, 30, 40, 64, 64, 30, 40, 64, 64
        d.KH, d.KW, d.stride, d.pad, d.flags, d.groups, d.dilation = 3, 3, 1, dil, 0, 1, dil
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = 64, 64 * 9 * 64, 64, 64
        d.tile, d.splitk, d.precision = 4, 1, 0
        L.check(L.lib().vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "dilated conv")
        assert (y.permute(0, 3, 1, 2).cpu() - ref).abs().max() < 2e-4, dil
    xa = S.normal01(4, "ap.x", (2, 128, 30, 40)).float()
    ya = torch.empty(2, 4, 5, 128, device="cuda")
    L.check(L.lib().vidc_avgpool2d(L.ptr(xa.permute(0, 2, 3, 1).contiguous().cuda()), L.ptr(ya), 2, 30, 40, 128, 128, 8, 8, 8, 8, 1, 0, 128,
                                   L.current_stream()), "avgpool")
    assert (ya.permute(0, 3, 1, 2).cpu() - F.avg_pool2d(xa, 8, stride=8, padding=(1, 0))).abs().max() < 1e-5


@gpu
def test_pipeline_without_gravity_uses_dorn(dorn_weights, seeded_weights):
    """RunDepthCompletion(use_gravity=False) (main.py:244-245, 270-271): DORN normals -> plane block -> depth completion, vs the
    oracle chain with the same RNG stream; RMSE 1e-3 like the gravity path."""
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    pipe = DepthCompletionPipeline(enriched_samples=200, use_gravity=False, rng=np.random.RandomState(5))
    pipe.load_state_dicts(dorn_weights, seeded_weights["dc"])
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    batch = S.synthetic_batch(1, 240, 320, 1234, frame0=5)
    got = pipe._call_cnn(batch).cpu()
    rng = np.random.RandomState(5)
    normals = O.dorn_forward(dorn_weights, batch["image"])
    ids = torch.from_numpy(S.plane_id_map(240, 320).astype(np.int64))
    di = O.extract_plane_depth(normals[0], ids, batch["sparse_depth"][0, 0], batch["homogeneous_coordinates"][0], rng=rng)
    enriched = O.enrich_sparse_depth(batch["sparse_depth"], di[None, None], 200, rng=rng)
    want = O.depth_completion_forward(seeded_weights["dc"], batch["image"], normals, enriched)
    assert float((got - want).pow(2).mean().sqrt()) < 1e-3


@gpu
def test_run_interleaved_without_gravity_equals_sequential(dorn_weights, seeded_weights):
    """`run_interleaved` with use_gravity=False: the DORN network + plane block of frame t on one HIP stream beside the enrichment and
    depth network of frame t-1 on another (pipeline._run_interleaved_two_programs).  Same programs and the same order of random
    draws as back-to-back `_call_cnn` calls: every depth map bit-identical, also with a sparse-depth input dense enough for the
    > 300-point branch (main.py:75-78) and with enriched_samples = 0."""
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    pipe = DepthCompletionPipeline(enriched_samples=200, use_gravity=False)
    pipe.load_state_dicts(dorn_weights, seeded_weights["dc"])
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    frames = [S.synthetic_batch(1, 240, 320, 1234, frame0=300 + i) for i in range(4)]
    g = torch.Generator().manual_seed(3)
    pix = torch.randperm(240 * 320, generator=g)[:4000]
    frames[2]["sparse_depth"] = frames[2]["sparse_depth"].clone()
    frames[2]["sparse_depth"].view(-1)[pix] = 1.0 + 3.0 * torch.rand(4000, generator=g)
    dev_frames = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in f.items()} for f in frames]
    for es in (200, 0):
        pipe.args.enriched_samples = es
        pipe.rng = np.random.RandomState(11)
        seq = [pipe._call_cnn(f).cpu() for f in dev_frames]
        pipe.rng = np.random.RandomState(11)
        il = [o.cpu() for o in pipe.run_interleaved(iter(dev_frames))]
        assert len(il) == len(seq) == 4
        for i, (a, b) in enumerate(zip(seq, il)):
            assert torch.equal(a, b), "frame %d differs (enriched_samples=%d)" % (i, es)
        assert not torch.equal(seq[0], seq[1])
    pipe.rng = np.random.RandomState(11)
    pipe.args.enriched_samples = 200
    host = [o.cpu() for o in pipe.run_interleaved(iter(frames), copy_outputs=False)]       # host-resident batches, program-owned outputs
    pipe.rng = np.random.RandomState(11)
    again = [pipe._call_cnn(f).cpu() for f in dev_frames]
    assert all(torch.equal(a, b) for a, b in zip(again, host))
