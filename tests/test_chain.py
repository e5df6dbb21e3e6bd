"""Persistent conv chain (csrc/conv_mfma.hip conv_chain_kernel, engine.Program._fuse_chains): a run of small convs as ONE launch.
CPU: the fusion pass on the recorded networks (which ops become chains, segment cuts survive).  GPU: a chain computes bit for bit what
the stand-alone kernel computes on the same tiling (64x64k2d4, split-K 1) -- same arithmetic, only the hand-off between layers
differs -- on a ResNet-layer3-shaped stack, repeated many times with changing inputs so that a stale L1/L2 line or a missed
dependency would show; and the whole networks stay within the usual bars with chains on."""
import os

import numpy as np
import pytest
import torch

from vi_depth_completion_amd import synthetic as S

torch.set_grad_enabled(False)


def _toy(monkeypatch, chain, precision, B=1, G=4, n_blocks=4, H=16, W=20, force_tile=True):
    """G-group stack of `n_blocks` Bottlenecks (1x1 1024->256, 3x3, 1x1 256->1024 + residual) at HxW, recorded as a Program."""
    from vi_depth_completion_amd import engine
    monkeypatch.setenv("VIDC_CHAIN", "1" if chain else "0")
    monkeypatch.setenv("VIDC_PRECISION", "mixed" if precision else "fp32")
    if force_tile and not chain:
        monkeypatch.setenv("VIDC_FORCE_TILE", str(engine.CHAIN_TILE))
    else:
        monkeypatch.delenv("VIDC_FORCE_TILE", raising=False)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for g in range(G):
                for i in range(n_blocks):
                    p = "g%d.b%d." % (g, i)
                    for name, (co, ci, k) in {"conv1": (256, 1024, 1), "conv2": (256, 256, 3), "conv3": (1024, 256, 1)}.items():
                        self.register_parameter((p + name + ".weight").replace(".", "_"), None)
            self.sd = {}

        def state_dict(self):
            return self.sd

    m = M()
    for g in range(G):
        for i in range(n_blocks):
            p = "g%d.b%d." % (g, i)
            for name, (co, ci, k) in {"conv1": (256, 1024, 1), "conv2": (256, 256, 3), "conv3": (1024, 256, 1)}.items():
                m.sd[p + name + ".weight"] = (S.normal01(5, p + name, (co, ci, k, k)).float() * (1.0 / (ci * k * k)) ** 0.5).cuda()
                m.sd[p + name + ".bias"] = (0.1 * S.normal01(6, p + name, (co,)).float()).cuda()
    ws = engine.WeightStore(m)
    prog = engine.Program(ws, torch.device("cuda"), B)
    x = prog.nhwc(H, W, 1024, G)
    prog.pinned.add(x.buf)
    prog.inputs["x"] = x
    cur = x
    for i in range(n_blocks):
        keys = engine.K(["g%d.b%d." % (g, i) for g in range(G)])
        a = prog.conv(cur, keys + "conv1", relu=True)
        b = prog.conv(a, keys + "conv2", relu=True, padding=1)
        cur = prog.conv(b, keys + "conv3", residual=cur, relu_after_residual=True)
    prog.mark_output("y", cur)
    prog.finalize()
    return prog


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("B,H,W", [(1, 16, 20), (1, 15, 20), (2, 15, 20)])
def test_chain_is_bit_identical_to_separate_launches(monkeypatch, precision, B, H, W):
    ref = _toy(monkeypatch, False, precision, B=B, H=H, W=W)
    got = _toy(monkeypatch, True, precision, B=B, H=H, W=W)
    assert got.n_chains == 1 and sum(1 for n in got.op_names if n.startswith("chain:")) == 1
    assert not any(n.startswith("chain:") for n in ref.op_names)
    xin_r, xin_g = ref.tensor(ref.inputs["x"]), got.tensor(got.inputs["x"])
    for it in range(25):                      # fresh inputs every time: buffers are rewritten, so stale cache lines would show
        x = S.normal01(100 + it, "chain.x", tuple(xin_r.shape)).float().cuda()
        xin_r.copy_(x)
        xin_g.copy_(x)
        ref.run()
        got.run()
        got.check_chains()
        yr, yg = ref.tensor(ref.outputs["y"]), got.tensor(got.outputs["y"])
        assert torch.isfinite(yr).all() and float(yr.abs().mean()) > 1e-3
        assert torch.equal(yr, yg), "iteration %d: max |diff| %.3e" % (it, float((yr - yg).abs().max()))


@pytest.mark.gpu
def test_chain_under_graph_replay_and_concurrent_streams(monkeypatch):
    """The captured chain (counter reset = memset node + kernel) replays correctly back to back, and two chains running at the same
    time on different streams both finish (nothing in the kernel needs every workgroup resident)."""
    ref = _toy(monkeypatch, False, 1)
    a = _toy(monkeypatch, True, 1)
    b = _toy(monkeypatch, True, 1)
    x = S.normal01(7, "chain.x2", tuple(ref.tensor(ref.inputs["x"]).shape)).float().cuda()
    for p in (ref, a, b):
        p.tensor(p.inputs["x"]).copy_(x)
    ref.run()
    want = ref.tensor(ref.outputs["y"]).clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        a.run()
        a.capture()
    with torch.cuda.stream(sb):
        b.run()
        b.capture()
    torch.cuda.synchronize()
    for _ in range(10):
        a.tensor(a.outputs["y"]).zero_()
        b.tensor(b.outputs["y"]).zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            a.launch()
            a.launch()
        with torch.cuda.stream(sb):
            b.launch()
        torch.cuda.synchronize()
        a.check_chains()
        b.check_chains()
        assert torch.equal(a.tensor(a.outputs["y"]), want) and torch.equal(b.tensor(b.outputs["y"]), want)


def test_fusion_pass_on_the_recorded_networks(monkeypatch):
    """CPU (dry run): with chains on, the frame program's layer3 bottlenecks 1..22 become one chain op; segment cuts survive; with
    VIDC_CHAIN=0 no chain op exists."""
    from vi_depth_completion_amd import pipeline
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval()
    dc = ModifiedFPN().eval()
    monkeypatch.setenv("VIDC_PRECISION", "mixed")
    monkeypatch.setenv("VIDC_CHAIN", "0")
    p0 = pipeline.build_frame_program(sn, dc, 1, 240, 320, torch.device("cpu"), dry_run=True)
    monkeypatch.setenv("VIDC_CHAIN", "1")
    p1 = pipeline.build_frame_program(sn, dc, 1, 240, 320, torch.device("cpu"), dry_run=True)
    assert p0.n_chains == 0 and p1.n_chains >= 1
    chains = [n for n in p1.op_names if n.startswith("chain:")]
    n_in_chains = sum(int(n.split(":")[1]) for n in chains)
    assert len(p1.op_names) == len(p0.op_names) - n_in_chains + len(chains)
    assert any(int(n.split(":")[1]) >= 66 for n in chains), chains            # layer3's 22 identical bottlenecks
    assert len(p1.cuts) == len(p0.cuts) == 1 and p1.flops == p0.flops
    # the two segments hold the same non-chain ops as before
    s0, s1 = p0.segments(), p1.segments()
    assert p0.op_names[s0[1][0]] == p1.op_names[s1[1][0]]
