"""`torch.ops.vidc.*` (vi_depth_completion_amd/torch_ops.py): registration and shape functions on CPU, numerics on the GPU
against torch-CPU fp32 / the oracle (same tolerances as tests/test_hip_parity.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vidc_oracle as O
from vi_depth_completion_amd import synthetic as S
from vi_depth_completion_amd import torch_ops as T

torch.set_grad_enabled(False)
FX, FY, CX, CY = 202.0, 202.0, 159.93827, 119.938015


def test_ops_are_registered():
    for name in T.OPS:
        assert hasattr(torch.ops.vidc, name), name


def test_no_cpu_kernel():
    """The operators exist for the GPU dispatch key only: a CPU call must fail loudly, never compute."""
    x = torch.zeros(1, 3, 240, 320)
    g = torch.tensor([[0.0, 1.0, 0.0]])
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.vidc.warp2dof_fwd(x, g, g, FX, FY, CX, CY, False)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.vidc.maxpool3x3s2(torch.zeros(1, 8, 8, 32))


def test_shape_functions():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(2, 60, 80, 64)
        w = torch.empty(128, 64, 3, 3)
        s = torch.empty(128)
        assert torch.ops.vidc.conv2d_bn_act(x, w, s, s, 2, 1, True, 1).shape == (2, 30, 40, 128)
        assert torch.ops.vidc.conv3x3_winograd(x, w, s, s, 4, True, 0).shape == (2, 60, 80, 128)
        assert torch.ops.vidc.maxpool3x3s2(x).shape == (2, 30, 40, 64)
        assert torch.ops.vidc.upsample_bilinear_ac(x, 120, 160, False).shape == (2, 120, 160, 64)
        assert torch.ops.vidc.stem_conv3x3s2(torch.empty(2, 3, 240, 320), torch.empty(64, 3, 3, 3), True).shape == (2, 120, 160, 64)
        assert torch.ops.vidc.head_conv1x1_upsample(x, torch.empty(1, 64, 1, 1), torch.empty(1), 1, 240, 320, True).shape == (2, 1, 240, 320)
        h, y = torch.ops.vidc.warp2dof_fwd(torch.empty(2, 3, 240, 320), torch.empty(2, 3), torch.empty(2, 3), FX, FY, CX, CY, False)
        assert h.shape == (2, 3, 3) and y.shape == (2, 3, 240, 320)


def test_plane_op_shape_functions():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        n = torch.empty(2, 3, 240, 320)
        ids = torch.empty(2, 240, 320, dtype=torch.uint8)
        slots = torch.empty(5, 4, dtype=torch.int32)
        mask, counts, scratch = torch.ops.vidc.plane_ransac_normal(n, ids, slots, torch.empty(1500, dtype=torch.int32))
        assert mask.shape == (5, 76800) and mask.dtype == torch.uint8 and counts.shape == (5, 300) and counts.dtype == torch.int32
        rec, sc = torch.ops.vidc.plane_offset(torch.empty(2, 240, 320, 3), torch.empty(2, 1, 240, 320), slots, mask, counts, scratch)
        assert rec.shape == (5, 16) and sc.shape == scratch.shape
        pd, rec2 = torch.ops.vidc.plane_project_depth(torch.empty(2, 240, 320, 3), slots, mask, sc, rec, torch.empty(2, 1, 240, 320))
        assert pd.shape == (2, 1, 240, 320) and rec2.shape == (5, 16)
        pd2, info = torch.ops.vidc.plane_finalize(torch.empty(2, 1, 240, 320), pd, rec2)
        assert info.shape == (2 * 300 + 1,) and info.dtype == torch.int32
        en = torch.ops.vidc.enrich_scatter(pd2, torch.empty(2, 1, 240, 320), torch.empty(7, dtype=torch.int32), torch.empty(3, dtype=torch.int32),
                                           torch.empty(600, dtype=torch.int32))
        assert en.shape == (2, 1, 240, 320)


@pytest.mark.gpu
def test_plane_ops_compose_to_the_plane_block():
    """torch.ops.vidc.plane_* / enrich_scatter chained by hand, with the host draws of plane.draw_*: the same arithmetic as
    plane.PlaneBlock (bit-identical), which test_hip_parity.py holds against the oracle and the reference's golden frames."""
    from vi_depth_completion_amd import plane
    b = S.synthetic_batch(2, 240, 320, 1234, frame0=3)
    nrm = F.normalize(S.normal01(5, "ops.plane.n", (2, 3, 240, 320)).float() * 0.05 + torch.tensor([0.0, -0.8, -0.6]).view(1, 3, 1, 1), dim=1).cuda()
    ids_np = [S.plane_id_map(240, 320)] * 2
    ds, homo = b["sparse_depth"].cuda(), b["homogeneous_coordinates"].cuda()
    pb = plane.PlaneBlock()
    di_ref, info_ref = pb.plane_depth(nrm, ids_np, ds, homo, rng=np.random.RandomState(9))
    di_ref, info_ref = di_ref.clone(), info_ref.clone()
    en_ref = pb.enrich(ds, di_ref, info_ref, 200, rng=np.random.RandomState(10)).clone()
    slots, hyp, _ = plane.draw_normal_hypotheses(ids_np, np.random.RandomState(9))
    ids = torch.from_numpy(np.stack(ids_np)).cuda()
    slots_d, hyp_d = torch.from_numpy(slots).cuda(), torch.from_numpy(hyp).cuda()
    mask, counts, scratch = torch.ops.vidc.plane_ransac_normal(nrm, ids, slots_d, hyp_d)
    rec, scratch = torch.ops.vidc.plane_offset(homo, ds, slots_d, mask, counts, scratch)
    pd, rec = torch.ops.vidc.plane_project_depth(homo, slots_d, mask, scratch, rec, ds)
    pd, info = torch.ops.vidc.plane_finalize(ds, pd, rec)
    assert torch.equal(pd, di_ref) and torch.equal(info, info_ref) and torch.equal(rec, pb.last_records)
    chunks = info.cpu().numpy()[:-1].reshape(2, -1)
    sub, offs = plane.draw_enrichment(chunks.sum(axis=1), 200, np.random.RandomState(10))
    base = (np.cumsum(chunks, axis=1) - chunks).astype(np.int32)
    en = torch.ops.vidc.enrich_scatter(pd, ds, torch.from_numpy(sub).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(base.reshape(-1)).cuda())
    assert torch.equal(en, en_ref)
    assert int((en != ds).sum()) > 100


@pytest.mark.gpu
def test_warp_ops_match_oracle():
    b = S.synthetic_batch(2, 240, 320, 1234)
    img, g, a = b["image"], b["gravity"], b["aligned_direction"]
    intr = O.Intrinsics(FX, FY, CX, CY)
    H_or, y_or = O.warp_forward(img, g, a, intr, False)
    H, y = torch.ops.vidc.warp2dof_fwd(img.cuda(), g.cuda(), a.cuda(), FX, FY, CX, CY, False)
    assert (H.cpu() - H_or).abs().max() <= 2e-5 * H_or.abs().max()
    d = (y.cpu() - y_or).abs()
    assert d.max() < 1e-3 and d.mean() < 2e-5
    nmap = S.normal01(1234, "ops.normalmap", (2, 3, 240, 320)).float()
    _, z_or = O.warp_inverse_normals(nmap, g, a, intr, False)
    z_or = F.normalize(z_or, dim=1)
    _, z = torch.ops.vidc.warp2dof_inv_rot_norm(nmap.cuda(), g.cuda(), a.cuda(), FX, FY, CX, CY, False, True)
    dz = (z.cpu() - z_or).abs()
    assert dz.mean() < 6e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [(0, 2e-4), (1, 2e-4)])
def test_conv_op_matches_torch(precision, tol):
    x = S.normal01(7, "ops.x", (2, 64, 30, 40)).float()
    w = S.normal01(7, "ops.w", (96, 64, 3, 3)).float() * (2.0 / (64 * 9)) ** 0.5
    scale = 0.5 + S.uniform01(7, "ops.s", (96,)).float()
    shift = 0.1 * S.normal01(7, "ops.b", (96,)).float()
    ref = F.relu(F.conv2d(x, w, stride=2, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    y = torch.ops.vidc.conv2d_bn_act(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), scale.cuda(), shift.cuda(), 2, 1, True, precision)
    assert y.shape == (2, 15, 20, 96)
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("precision,tol", [(0, 2e-4), (1, 2e-3)])
def test_winograd_conv_op_matches_torch(m, precision, tol):
    x = S.normal01(8, "ops.x", (2, 64, 30, 41)).float()
    w = S.normal01(8, "ops.w", (96, 64, 3, 3)).float() * (2.0 / (64 * 9)) ** 0.5
    scale = 0.5 + S.uniform01(8, "ops.s", (96,)).float()
    shift = 0.1 * S.normal01(8, "ops.b", (96,)).float()
    ref = F.relu(F.conv2d(x, w, stride=1, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    y = torch.ops.vidc.conv3x3_winograd(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), scale.cuda(), shift.cuda(), m, True, precision)
    assert y.shape == (2, 30, 41, 96)
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max() < tol


@pytest.mark.gpu
def test_glue_ops_match_torch():
    x = S.normal01(9, "ops.g", (2, 64, 31, 41)).float()
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    mp = torch.ops.vidc.maxpool3x3s2(xh).cpu().permute(0, 3, 1, 2)
    assert torch.equal(mp, F.max_pool2d(x, 3, 2, 1))
    up = torch.ops.vidc.upsample_bilinear_ac(xh, 62, 82, True).cpu().permute(0, 3, 1, 2)
    ref = F.relu(F.interpolate(x, size=(62, 82), mode="bilinear", align_corners=True))
    assert (up - ref).abs().max() < 1e-5
    img = S.uniform01(9, "ops.img", (2, 3, 240, 320))
    w = S.normal01(9, "ops.sw", (64, 3, 3, 3)).float() * 0.2
    st = torch.ops.vidc.stem_conv3x3s2(img.cuda(), w.cuda(), True).cpu().permute(0, 3, 1, 2)
    assert (st - F.relu(F.conv2d(img, w, stride=2, padding=1))).abs().max() < 1e-5
    hw = S.normal01(9, "ops.hw", (1, 64, 1, 1)).float() * 0.1
    hb = torch.tensor([0.3])
    hd = torch.ops.vidc.head_conv1x1_upsample(xh, hw.cuda(), hb.cuda(), 1, 120, 160, True).cpu()
    ref = F.relu(F.interpolate(F.conv2d(x, hw, hb, padding=1), size=(120, 160), mode="bilinear", align_corners=True))
    assert (hd - ref).abs().max() < 2e-5


@pytest.mark.gpu
def test_raw_stream_accessor_follows_the_current_stream():
    """`_lib.current_stream()` (the hipStream_t every C entry is handed) reads torch's current stream through the private raw accessor when
    this torch has it: it must name the same stream as `torch.cuda.current_stream()`, on the default stream and inside stream contexts."""
    import torch
    from vi_depth_completion_amd import _lib as L
    assert L.current_stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert L.current_stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
        inner = torch.cuda.Stream()
        with torch.cuda.stream(inner):
            assert L.current_stream() == inner.cuda_stream
        assert L.current_stream() == side.cuda_stream
    assert L.current_stream() == torch.cuda.current_stream().cuda_stream
