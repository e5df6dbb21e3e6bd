"""Winograd F(m x m, 3x3) path (csrc/winograd.hip + the grouped GEMM launch of csrc/conv_mfma.hip).

The reference's 3x3 layers are `nn.Conv2d(c, c, 3, 1, 1)` + BatchNorm2d + ReLU run by ATen (networks/surface_normal.py:75-141,
networks/depth_completion.py:77-143), so the pin is `F.conv2d` on the CPU; oracle/winograd_oracle.py restates the three transforms
so each kernel is also checked on its own.

Tolerances: a transform kernel vs its fp64-computed restatement: 4 ulp-ish of the largest term (1e-5 relative to max |value|); the
composed conv vs F.conv2d fp32: 2e-4 abs on O(1) data like the direct kernel's tests (fp32 mode), 2e-3 in the bf16x3 mode (its products
carry 2^-16 relative error; the direct bf16x3 tests use the same bound).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import winograd_oracle as WO
from vi_depth_completion_amd import synthetic as S

torch.set_grad_enabled(False)
DEV = "cuda"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _case(seed, B, H, W, cin, cout, G):
    x = S.normal01(seed, "x", (B, G * cin, H, W)).float()
    w = [S.normal01(seed, "w%d" % g, (cout, cin, 3, 3), scale=float(np.sqrt(2.0 / (cin * 9)))).float() for g in range(G)]
    s1 = S.uniform01(seed, "s1", (G, cout)) + 0.5
    b1 = S.normal01(seed, "b1", (G, cout)).float() * 0.1
    return x, w, s1, b1


def _ref(x, w, s1, b1, G):
    cin = x.shape[1] // G
    outs = [F.conv2d(x[:, g * cin:(g + 1) * cin], w[g], None, 1, 1) * s1[g].view(1, -1, 1, 1) + b1[g].view(1, -1, 1, 1) for g in range(G)]
    return torch.cat(outs, 1)


# ---- CPU: the restatement against F.conv2d, and the host-side recording ------------------------------------------------------------
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("shape", [(1, 9, 11, 32, 32, 1), (2, 16, 20, 64, 32, 3), (1, 15, 20, 32, 64, 2)])
def test_oracle_restatement_is_a_3x3_conv(m, shape):
    B, H, W, cin, cout, G = shape
    x, w, _s, _b = _case(3, B, H, W, cin, cout, G)
    y = WO.conv3x3(nhwc(x), w, m)
    ref = torch.cat([F.conv2d(x[:, g * cin:(g + 1) * cin], w[g], None, 1, 1) for g in range(G)], 1)
    assert (nchw(y) - ref).abs().max().item() < 5e-5


def test_winograd_choice_policy(monkeypatch):
    from vi_depth_completion_amd import engine as E
    monkeypatch.delenv("VIDC_WINOGRAD", raising=False)
    monkeypatch.setattr(E, "_TUNING", {})                   # the heuristic, not the measured per-layer verdicts
    assert E.winograd_choice(1, 60, 80, 768, 768, 3, 3, 1, 1, 1, 1, precision="fp32") == 4          # large map
    assert E.winograd_choice(1, 15, 20, 1536, 1536, 3, 3, 1, 1, 1, 2, precision="fp32") == 2        # small map
    assert E.winograd_choice(1, 60, 80, 64, 64, 3, 3, 1, 1, 1, 4, precision="fp32") == 0            # K = 64 per GEMM: the transforms would dominate
    # the mixed mode without a measured verdict stays direct (the table's verdict for 73 of 112 layers; a bf16x3 direct conv is cheap)
    assert E.winograd_choice(1, 60, 80, 768, 768, 3, 3, 1, 1, 1, 1, precision="mixed") == 0
    monkeypatch.setenv("VIDC_PRECISION", "fp32")
    assert E.winograd_choice(1, 60, 80, 128, 128, 3, 3, 2, 1, 1, 1) == 0          # stride 2
    assert E.winograd_choice(1, 60, 80, 256, 256, 3, 3, 1, 6, 6, 1) == 0          # dilated (DORN's ASPP)
    assert E.winograd_choice(1, 60, 80, 256, 256, 1, 1, 1, 0, 1, 1) == 0
    monkeypatch.setenv("VIDC_WINOGRAD", "0")
    assert E.winograd_choice(1, 60, 80, 768, 768, 3, 3, 1, 1, 1, 1) == 0
    monkeypatch.setenv("VIDC_WINOGRAD", "2")
    assert E.winograd_choice(1, 60, 80, 64, 64, 3, 3, 1, 1, 1, 4) == 2
    # 32-bit offsets of the GEMM kernel: a transform-domain tensor of 2 GiB or more falls back to the direct form
    assert E.winograd_choice(32, 720, 1280, 768, 768, 3, 3, 1, 1, 1, 1) == 0


def test_winograd_choice_verdict_5_is_for_its_own_group_count_only(monkeypatch):
    """Table verdict 5 = F(4 x 4) in ONE launch (csrc/wfused.hip): measured for one group count; programs that borrow the entry of another group
    count (stand-alone 1- / 3-group programs) get the three-launch F(4 x 4)."""
    from vi_depth_completion_amd import engine as E
    monkeypatch.delenv("VIDC_WINOGRAD", raising=False)
    monkeypatch.setenv("VIDC_TUNING_OVERRIDE", '{"W:M1280_N256_K2304_k3s1_G4": [5, 0]}')
    monkeypatch.setattr(E, "_TUNING", None)
    try:
        assert E.winograd_choice(4, 16, 20, 256, 256, 3, 3, 1, 1, 1, 4, precision="fp32") == 5
        assert E.winograd_choice(4, 16, 20, 256, 256, 3, 3, 1, 1, 1, 2, precision="fp32") == 4       # no entry of its own: borrows G4's, as three launches
        assert E.winograd_choice(4, 16, 20, 256, 256, 3, 3, 1, 1, 1, 4, precision="mixed") == 0
    finally:
        E._TUNING = None


def test_recorded_program_has_winograd_triples(monkeypatch):
    """Dry-run recording (no HIP call): every qualifying 3x3 conv becomes wino_in -> grouped 1x1 GEMM -> wino_out; executed FLOPs drop,
    the reference-formulation FLOPs do not."""
    import vi_depth_completion_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("libvidc.so not built")
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    monkeypatch.setenv("VIDC_PRECISION", "fp32")
    progs = {}
    for mode in ("0", "auto"):
        monkeypatch.setenv("VIDC_WINOGRAD", mode)
        sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval()
        progs[mode] = sn.build_program(1, torch.device("cpu"), dry_run=True)
    d, w = progs["0"], progs["auto"]
    kd, kw = [k for k, *_ in d.ops], [k for k, *_ in w.ops]
    n = kw.count("wino_in")
    assert n > 0 and kw.count("wino_out") == n and kd.count("wino_in") == 0
    assert kw.count("conv") == kd.count("conv")                       # one GEMM launch per replaced conv
    assert w.ref_flops == d.ref_flops and w.direct_flops == d.flops and w.flops < d.flops
    for i, (kind, _r, _w, kwargs) in enumerate(w.ops):
        if kind == "wino_in":
            assert w.ops[i + 1][0] == "conv" and w.ops[i + 1][3].get("wino") == kwargs["m"] and w.ops[i + 2][0] == "wino_out"
            a2 = (kwargs["m"] + 2) ** 2
            desc = w.c_ops[i + 1].u.conv
            assert desc.groups == a2 * len(w.ops[i + 1][3]["keys"]) and desc.KH == 1 and desc.flags == 0 and desc.p_gs == 0
            assert desc.ldx == a2 * kwargs["x"].C * kwargs["x"].G and w.c_ops[i].u.g.i[8] == desc.ldx and w.c_ops[i + 2].u.g.i[8] == desc.ldy


def test_recorded_program_with_one_launch_layers(monkeypatch):
    """Dry-run recording with VIDC_WINO_FUSED (no HIP call): a qualifying F(4 x 4) layer is ONE conv op on the fused tile -- the descriptor of the 3x3
    conv with the real affine, w_gs = 36 Cout Cin -- and no transform ops around it; executed / reference FLOPs are those of the three-launch form."""
    import vi_depth_completion_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("libvidc.so not built")
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    monkeypatch.setenv("VIDC_PRECISION", "fp32")
    monkeypatch.setenv("VIDC_WINOGRAD", "4")
    progs = {}
    for knob in ("0", "100000"):
        monkeypatch.setenv("VIDC_WINO_FUSED", knob)
        monkeypatch.setenv("VIDC_WINO_FUSED_MAXC", "4096")
        sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval()
        progs[knob] = sn.build_program(1, torch.device("cpu"), dry_run=True)
    t, f = progs["0"], progs["100000"]
    kt, kf = [k for k, *_ in t.ops], [k for k, *_ in f.ops]
    assert kt.count("wino_in") > 0 and kf.count("wino_in") == 0 and kf.count("wino_out") == 0
    assert kf.count("conv") == kt.count("conv") and len(kf) == len(kt) - 2 * kt.count("wino_in")
    assert (f.flops, f.ref_flops, f.direct_flops) == (t.flops, t.ref_flops, t.direct_flops)
    n = 0
    for i, (kind, _r, _w, kwargs) in enumerate(f.ops):
        if kind == "conv" and kwargs.get("wino_fused"):
            d = f.c_ops[i].u.conv
            assert d.tile == L.TILE_WINO4_FUSED and (d.KH, d.KW, d.stride, d.pad) == (3, 3, 1, 1) and d.precision == L.PREC_FP32
            assert d.w_gs == 36 * d.Cout * d.Cin and d.p_gs == d.Cout and not (d.flags & ~(L.RELU1 | L.AFFINE2 | L.RELU2))
            assert "@wino4f" in f.op_names[i]
            n += 1
    assert n == kt.count("wino_in")
    # the mixed mode never records it
    monkeypatch.setenv("VIDC_PRECISION", "mixed")
    sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval()
    m = sn.build_program(1, torch.device("cpu"), dry_run=True)
    assert not any(kw.get("wino_fused") for _k, _r, _w, kw in m.ops)


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------
WINO_SHAPES = [
    # B, H, W, cin, cout, G
    (1, 60, 80, 128, 128, 1),
    (1, 30, 40, 256, 256, 3),      # the decoders' level-2 launch (three branches grouped)
    (2, 15, 20, 256, 256, 2),      # odd height: a partial tile row with m = 2 and m = 4
    (1, 8, 10, 512, 256, 1),
    (1, 9, 11, 32, 64, 1),         # odd sizes, partial tiles on both edges
    (1, 64, 80, 64, 32, 4),
    (4, 64, 80, 256, 32, 2),       # large enough for the four-channels-per-thread form of the transforms with m = 2 and m = 4 (the shapes above
]                                  # run one channel per thread: csrc/winograd.hip vecn)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("shape", WINO_SHAPES)
def test_transforms_vs_restatement(m, shape):
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = shape
    x, w, s1, b1 = _case(21, B, H, W, cin, cout, G)
    xh = nhwc(x)
    v = ops.winograd_input_transform(xh.to(DEV), cin, m).cpu()
    v_ref = WO.input_transform(xh, cin, m)
    assert v.shape == v_ref.shape and (v - v_ref).abs().max().item() <= 1e-5 * v_ref.abs().max().item()
    u = ops.winograd_weight_transform(w[0].to(DEV), m).cpu()
    u_ref = WO.weight_transform(w[0], m)                                # fp64 inside, rounded once: equal up to the last fp32 bit
    assert (u - u_ref).abs().max().item() <= 2.0 ** -23 * u_ref.abs().max().item()
    a2 = (m + 2) ** 2
    mm = S.normal01(22, "mm", (v.shape[0], a2 * G * cout)).float()
    one, zero = torch.ones(G, cout), torch.zeros(G, cout)
    y = ops.winograd_output_transform(mm.to(DEV), B, H, W, cout, m, one.to(DEV), zero.to(DEV)).cpu()
    y_ref = WO.output_transform(mm, B, H, W, cout, m)
    assert y.shape == y_ref.shape and (y - y_ref).abs().max().item() <= 1e-5 * y_ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 4])
def test_transform_bits_do_not_depend_on_the_groups_sharing_the_launch(m):
    """A frame program's first / drain ticks run the transforms restricted to one resp. three of the four pyramids
    (engine.Program.group_variant): fewer channels per launch, hence the one-channel-per-thread instantiation where the full launch takes
    four.  Both must round alike (csrc/winograd.hip compiles without mul + add contraction for that reason): group 0 alone == group 0 of 4."""
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = (1 if m == 2 else 4), 64, 80, 256, 256, 4      # one group: B*th*tw*(C/4) = 81920 < 131072 <= 327680: four groups
    x = nhwc(S.normal01(31, "x", (B, G * cin, H, W)).float()).to(DEV)
    a2 = (m + 2) ** 2
    v4 = ops.winograd_input_transform(x, cin, m).view(-1, G, a2 * cin)
    v1 = ops.winograd_input_transform(x[..., :cin].contiguous(), cin, m).view(-1, 1, a2 * cin)
    assert torch.equal(v4[:, :1], v1)
    mm = S.normal01(32, "mm", (v4.shape[0], G, a2 * cout)).float().to(DEV)
    s1, b1 = S.uniform01(33, "s", (G, cout)).float().to(DEV) + 0.5, S.normal01(34, "b", (G, cout)).float().to(DEV)
    y4 = ops.winograd_output_transform(mm.view(mm.shape[0], -1), B, H, W, cout, m, s1, b1, relu1=True)
    y1 = ops.winograd_output_transform(mm[:, 0].contiguous(), B, H, W, cout, m, s1[:1], b1[:1], relu1=True)
    assert torch.equal(y4[..., :cout], y1)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("shape", WINO_SHAPES)
def test_winograd_conv_vs_torch(m, shape):
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = shape
    x, w, s1, b1 = _case(23, B, H, W, cin, cout, G)
    ref = F.relu(_ref(x, w, s1, b1, G))
    y = ops.conv3x3_winograd(nhwc(x).to(DEV), [t.to(DEV) for t in w], s1.to(DEV), b1.to(DEV), m, relu1=True)
    err = (nchw(y).cpu() - ref).abs().max().item()
    assert err < 2e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 4])
def test_winograd_conv_second_affine_and_direct_kernel_agree(m):
    """conv1_3's epilogue (affine, relu, affine, relu: surface_normal.py:41-43) through the output transform; and the Winograd result
    against the direct MFMA kernel on the same operands (both fp32: they differ by summation order only)."""
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = 1, 30, 40, 256, 128, 2
    x, w, s1, b1 = _case(25, B, H, W, cin, cout, G)
    s2 = S.uniform01(25, "s2", (G, cout)) + 0.5
    b2 = S.normal01(25, "b2", (G, cout)).float() * 0.1
    t = F.relu(_ref(x, w, s1, b1, G))
    ref = F.relu(t * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1))
    xd = nhwc(x).to(DEV)
    y = ops.conv3x3_winograd(xd, [t_.to(DEV) for t_ in w], s1.to(DEV), b1.to(DEV), m, relu1=True, scale2=s2.to(DEV), shift2=b2.to(DEV), relu2=True)
    assert (nchw(y).cpu() - ref).abs().max().item() < 2e-4
    wp = torch.stack([ops.pack_conv_weight(wg.to(DEV)) for wg in w])
    yd = ops.conv2d_bn_act(xd, wp, s1.to(DEV), b1.to(DEV), 3, 3, 1, 1, relu1=True, scale2=s2.to(DEV), shift2=b2.to(DEV), relu2=True, groups=G)
    assert (y - yd).abs().max().item() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("shape", [WINO_SHAPES[1], WINO_SHAPES[2]])
def test_winograd_conv_bf16x3_and_split_output(m, shape):
    """Mixed mode: V is written as the split-bf16 image by the input transform, U packed split, GEMMs as 3-pass bf16 MFMA; the output
    transform also writes the split image of its result for a following bf16x3 conv (and may skip the fp32 store)."""
    from vi_depth_completion_amd import ops
    import vi_depth_completion_amd._lib as L
    B, H, W, cin, cout, G = shape
    x, w, s1, b1 = _case(27, B, H, W, cin, cout, G)
    ref = F.relu(_ref(x, w, s1, b1, G))
    xd = nhwc(x).to(DEV)
    sp = torch.zeros((B, H, W, G * cout), dtype=torch.float32, device=DEV)
    y = ops.conv3x3_winograd(xd, [t.to(DEV) for t in w], s1.to(DEV), b1.to(DEV), m, relu1=True, precision=L.PREC_BF16X3, split_out=sp)
    assert (nchw(y).cpu() - ref).abs().max().item() < 2e-3
    assert torch.equal(sp, ops.split_bf16x3(y))
    sp2 = torch.zeros_like(sp)
    ops.conv3x3_winograd(xd, [t.to(DEV) for t in w], s1.to(DEV), b1.to(DEV), m, relu1=True, precision=L.PREC_BF16X3, split_out=sp2, no_f32_out=True)
    assert torch.equal(sp2, sp)


@pytest.mark.gpu
def test_transforms_reject_bad_arguments():
    import vi_depth_completion_amd._lib as L
    lib = L.lib()
    x = torch.zeros(1, 8, 8, 32, device=DEV)
    v = torch.zeros(16, 16 * 32, device=DEV)
    st = L.current_stream()
    assert lib.vidc_winograd_input_transform(L.ptr(x), L.ptr(v), 1, 8, 8, 32, 32, 32, 3, 0, 0, st) == -2        # m = 3
    assert lib.vidc_winograd_input_transform(None, L.ptr(v), 1, 8, 8, 32, 32, 32, 2, 0, 0, st) == -1
    assert lib.vidc_winograd_input_transform(L.ptr(x), L.ptr(v), 1, 8, 8, 32, 32, 24, 2, 0, 0, st) == -2       # C not a multiple of Cin
    assert lib.vidc_winograd_input_transform(L.ptr(x), L.ptr(v), 1, 8, 8, 32, 32, 32, 2, 0, 64, st) == -2      # row stride below a*a*C
    one = torch.ones(32, device=DEV)
    assert lib.vidc_winograd_output_transform(L.ptr(v), L.ptr(x), None, L.ptr(one), L.ptr(one), None, None, 1, 8, 8, 32, 32, 32, 2, L.RESIDUAL, 0, st) == -2
    assert lib.vidc_winograd_output_transform(L.ptr(v), L.ptr(x), None, L.ptr(one), L.ptr(one), None, None, 1, 8, 8, 32, 32, 32, 2, L.AFFINE2, 0, st) == -1
    assert lib.vidc_winograd_output_transform(L.ptr(v), None, None, L.ptr(one), L.ptr(one), None, None, 1, 8, 8, 32, 32, 32, 2, 0, 0, st) == -1
    assert b"winograd" in lib.vidc_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_networks_winograd_vs_direct(seeded_weights, precision, monkeypatch):
    """The whole path with and without the Winograd layers on one synthetic 320x240 frame: same depth map up to fp32 rounding
    (oracle-side experiment: RMSE 1e-6; here bound at 2e-5 fp32 / 1e-4 mixed, bar 1e-3), and both within the bar of the oracle."""
    from oracle import vidc_oracle as O
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    monkeypatch.setenv("VIDC_PRECISION", precision)
    ids = S.plane_id_map(240, 320)
    batch = S.synthetic_batch(1, 240, 320, 1234)
    dev_batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    out = {}
    for mode in ("0", "auto", "2"):
        monkeypatch.setenv("VIDC_WINOGRAD", mode)
        pipe = DepthCompletionPipeline(enriched_samples=200, device=torch.device(DEV), rng=np.random.RandomState(7))
        pipe.load_state_dicts({k: v.to(DEV) for k, v in seeded_weights["sn"].items()}, {k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
        pipe.plane_masks_extraction = FixedPlaneMask(ids)
        out[mode] = pipe._call_cnn(dev_batch).cpu()
        del pipe
    wp_intr = O.Intrinsics(202.0, 202.0, 159.93827, 119.938015)
    ref = O.call_cnn(seeded_weights["sn"], seeded_weights["dc"], batch, [ids], wp_intr, 200, rng=np.random.RandomState(7))
    bound = 2e-5 if precision == "fp32" else 1e-4
    for mode in ("auto", "2"):
        assert float((out[mode] - out["0"]).pow(2).mean().sqrt()) < bound
        assert not torch.equal(out[mode], out["0"])             # (the Winograd layers did run)
    for mode in out:
        assert float((out[mode] - ref).pow(2).mean().sqrt()) < (1e-4 if precision == "fp32" else 1e-3)
