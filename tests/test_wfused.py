"""Winograd F(4x4, 3x3) in one launch (csrc/wfused.hip, tile VIDC_TILE_WINO4_FUSED): the 3x3 / stride 1 / pad 1 Conv2d + BatchNorm2d + ReLU layers of
the small maps (the 22 conv2 of ResNet-101 layer 3, torchvision Bottleneck as used at networks/surface_normal.py:27-35,48-50) without the V / M tensors.
Against F.conv2d in float64; against the three-launch Winograd path (same arithmetic, another summation order); bits independent of the batch a tile
is part of and of the groups sharing the launch (what engine.Program.group_variant and the stream mode rely on)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vi_depth_completion_amd import synthetic as S

pytestmark = pytest.mark.gpu
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


def _ref64(x, w, s1, b1, G):
    cin = x.shape[1] // G
    outs = [F.conv2d(x[:, g * cin:(g + 1) * cin].double(), w[g].double(), None, 1, 1) * s1[g].double().view(1, -1, 1, 1) + b1[g].double().view(1, -1, 1, 1) for g in range(G)]
    return torch.cat(outs, 1)


SHAPES = [(4, 16, 20, 256, 256, 4),      # layer 3 at the program batch of the timed configuration: 80 tiles per group
          (4, 32, 40, 128, 128, 4),      # layer 2
          (1, 16, 20, 256, 256, 1),      # one frame, one group: 20 tiles (a block of 16 and a ragged one)
          (3, 15, 18, 64, 96, 2),        # ragged map (tiles cut by both borders), Cin = 64, Cout not a power of two
          (2, 8, 10, 512, 64, 1),
          (1, 5, 7, 32, 32, 3)]          # a single K chunk


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_vs_float64_and_three_launch_path(shape):
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = shape
    x, w, s1, b1 = _case(41, B, H, W, cin, cout, G)
    ref = F.relu(_ref64(x, w, s1, b1, G))
    xd, wd = nhwc(x).to(DEV), [t.to(DEV) for t in w]
    y = ops.conv3x3_winograd_fused(xd, wd, s1.to(DEV), b1.to(DEV), relu1=True)
    torch.cuda.synchronize()
    err = (nchw(y).cpu().double() - ref).abs().max().item()
    assert err < 2e-4, err
    y3 = ops.conv3x3_winograd(xd, wd, s1.to(DEV), b1.to(DEV), 4, relu1=True)
    assert (y - y3).abs().max().item() < 2e-4
    e3 = (nchw(y3).cpu().double() - ref).abs().max().item()
    assert err < 5.0 * e3 + 1e-6, (err, e3)          # the same arithmetic in another summation order (one accumulator per position over all of K)


def test_second_affine_and_no_relu():
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = 2, 16, 20, 128, 64, 2
    x, w, s1, b1 = _case(43, B, H, W, cin, cout, G)
    s2 = S.uniform01(43, "s2", (G, cout)) + 0.5
    b2 = S.normal01(43, "b2", (G, cout)).float() * 0.1
    xd, wd = nhwc(x).to(DEV), [t.to(DEV) for t in w]
    t = F.relu(_ref64(x, w, s1, b1, G))
    ref = F.relu(t * s2.double().view(1, -1, 1, 1) + b2.double().view(1, -1, 1, 1))
    y = ops.conv3x3_winograd_fused(xd, wd, s1.to(DEV), b1.to(DEV), relu1=True, scale2=s2.to(DEV), shift2=b2.to(DEV), relu2=True)
    assert (nchw(y).cpu().double() - ref).abs().max().item() < 2e-4
    y = ops.conv3x3_winograd_fused(xd, wd, s1.to(DEV), b1.to(DEV))
    assert (nchw(y).cpu().double() - _ref64(x, w, s1, b1, G)).abs().max().item() < 2e-4
    assert bool((y < 0).any())


def test_bits_do_not_depend_on_batch_partners_or_groups_in_the_launch():
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = 4, 16, 20, 256, 256, 4
    x, w, s1, b1 = _case(45, B, H, W, cin, cout, G)
    xd, wd = nhwc(x).to(DEV), [t.to(DEV) for t in w]
    full = ops.conv3x3_winograd_fused(xd, wd, s1.to(DEV), b1.to(DEV), relu1=True)
    for b in range(B):          # one frame alone: other tile blocks, other workgroup ids
        one = ops.conv3x3_winograd_fused(xd[b:b + 1].contiguous(), wd, s1.to(DEV), b1.to(DEV), relu1=True)
        assert torch.equal(one[0], full[b]), b
    # groups 1..3 alone (pointer offsets in engine.Program.group_variant; here: slices)
    part = ops.conv3x3_winograd_fused(xd[..., cin:].contiguous(), wd[1:], s1[1:].to(DEV), b1[1:].to(DEV), relu1=True)
    assert torch.equal(part, full[..., cout:])
    again = ops.conv3x3_winograd_fused(xd, wd, s1.to(DEV), b1.to(DEV), relu1=True)
    assert torch.equal(again, full)


def test_magnitudes_activations_1e3_weights_1e_3():
    """Real checkpoints are unavailable: F(4x4) amplifies operands by up to 100 (B^T) / 1/24 (G); activations x 1e3 and weights x 1e-3 must stay
    within the fp32 bar RELATIVE to the result's scale."""
    from vi_depth_completion_amd import ops
    B, H, W, cin, cout, G = 4, 16, 20, 256, 256, 1
    x, w, s1, b1 = _case(47, B, H, W, cin, cout, G)
    x, w = x * 1e3, [t * 1e-3 for t in w]
    ref = _ref64(x, w, s1, b1 * 0, G)
    y = ops.conv3x3_winograd_fused(nhwc(x).to(DEV), [t.to(DEV) for t in w], s1.to(DEV), (b1 * 0).to(DEV))
    rel = (nchw(y).cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 2e-5, rel


def test_refusals():
    from vi_depth_completion_amd import ops
    x, w, s1, b1 = _case(49, 1, 8, 8, 48, 32, 1)          # Cin % 32 != 0
    with pytest.raises(RuntimeError, match="Cin"):
        ops.conv3x3_winograd_fused(nhwc(x).to(DEV), [t.to(DEV) for t in w], s1.to(DEV), b1.to(DEV))


def test_engine_records_one_launch_per_layer_and_the_stream_keeps_its_properties(seeded_weights, monkeypatch):
    """Where the measured table says 5 (or behind VIDC_WINO_FUSED) the engine records an F(4x4) layer as ONE conv op on the fused tile (no wino_in /
    wino_out around it);
    the whole path stays within the fp32 bar of the default recording, and an item's depth map is bit-identical whatever partner, slot or lane count it
    had -- including the first / drain ticks, which run the layer with one and three of its four groups (engine.Program.group_variant)."""
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    monkeypatch.setenv("VIDC_PRECISION", "fp32")
    frames = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=500 + i).items()} for i in range(7)]
    rng_of = lambda f: np.random.RandomState(7000 + f)      # noqa: E731

    def pipe():
        p = DepthCompletionPipeline(enriched_samples=200)
        p.load_state_dicts(seeded_weights["sn"], seeded_weights["dc"])
        p.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
        return p

    def run(p, first, last, lanes, F=4):
        return [o.cpu() for o in p.run_interleaved(iter(frames[first:last]), lanes=lanes, frames_per_launch=F, frame_rng=lambda i: rng_of(first + i))]

    from vi_depth_completion_amd import engine as E
    monkeypatch.setenv("VIDC_WINO_FUSED", "0")               # the three launches everywhere, whatever the table says
    base = run(pipe(), 0, 7, 1)
    # the way a measured table adopts it: verdict 5 (= F(4 x 4) in one launch) for the layer-3 conv2 of the four-pyramid launch at program batch 4
    monkeypatch.delenv("VIDC_WINO_FUSED", raising=False)
    monkeypatch.setenv("VIDC_TUNING_OVERRIDE", '{"W:M1200_N256_K2304_k3s1_G4": [5, 0]}')      # (240 x 320 frames: 15 x 20 maps in layer 3)
    monkeypatch.setattr(E, "_TUNING", None)
    p = pipe()
    ref = run(p, 0, 7, 1)
    names = p.frame_program(4, 240, 320).op_names          # (lane 0 runs the pipeline's own cached program of that batch)
    assert sum("@wino4f" in n for n in names) == 22, "the layer-3 conv2 layers were not recorded on the fused tile"
    assert sum(n.startswith("wino_in") for n in names) == sum("@wino" in n and "@wino4f" not in n for n in names)      # transforms only around the other Winograd layers
    monkeypatch.setattr(E, "_TUNING", None)                # (later tests read the committed table again)
    for f, (a, b) in enumerate(zip(base, ref)):
        assert float((a - b).pow(2).mean().sqrt()) < 2e-5, f
    assert any(not torch.equal(a, b) for a, b in zip(base, ref)), "the fused layers did not run"
    for lanes in (2, 3):
        got = run(p, 0, 7, lanes)
        for f, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), "frame %d differs with %d lanes" % (f, lanes)
    got = run(p, 1, 7, 2)
    for f, b in zip(range(1, 7), got):
        assert torch.equal(ref[f], b), "frame %d differs when grouped with other frames" % f
