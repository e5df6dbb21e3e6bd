"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/vidc.h declares;
host-side logic that needs no GPU (conv planning, RNG draws, program recording)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vi_depth_completion_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    L.build()
    return L.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "vidc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vidc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vidc_version() == 1
    assert lib.vidc_last_error() is not None
    # ... and NOTHING else: the library is linked with -fvisibility=hidden and csrc/vidc.map, so template instantiations, kernel handles and
    # the per-TU hip symbols stay local (VERDICT r5 hygiene: `_ZNSt6vectorI7vidc_op...` used to be a dynamic symbol)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "vi_depth_completion_amd", "libvidc.so")], capture_output=True, text=True, check=True)
    exported = {ln.split()[-1] for ln in nm.stdout.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_struct_layout_matches_header():
    # vidc_conv_desc: 9 pointers, 16 int32, 5 int64, 4 int32, 1 pointer  (natural alignment)
    assert C.sizeof(L.ConvDesc) == 9 * 8 + 16 * 4 + 5 * 8 + 4 * 4 + 8
    assert C.sizeof(L.GenericArgs) == 8 * 8 + 16 * 4 + 8 * 4
    assert C.sizeof(L.Op) == 16 + max(C.sizeof(L.ConvDesc), C.sizeof(L.GenericArgs))


def test_error_reporting_without_gpu(lib):
    assert lib.vidc_conv2d_bn_act(None, None) == -1
    assert b"null" in lib.vidc_last_error()
    d = L.ConvDesc()
    d.x = d.w = d.y = d.scale1 = d.shift1 = 8
    d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout, d.ldy = 1, 8, 8, 48, 48, 8, 8, 64, 64
    d.KH = d.KW = d.stride = d.groups = d.splitk = 1
    assert lib.vidc_conv2d_bn_act(C.byref(d), None) == -2
    assert b"Cin" in lib.vidc_last_error()
    # Cout must be whole 32-channel MFMA tiles (the stores are guarded per tile, not per lane): 48 and 8 are refused up front
    for cout in (48, 8):
        d.Cin, d.ldx, d.Cout, d.ldy = 64, 64, cout, 64
        assert lib.vidc_conv2d_bn_act(C.byref(d), None) == -2
        assert b"Cout" in lib.vidc_last_error()
    # one group's weights must stay addressable with 32-bit buffer offsets
    d.Cout, d.ldy, d.Cin, d.ldx, d.KH, d.KW, d.pad = 8192, 8192, 8192, 8192, 3, 3, 1
    assert lib.vidc_conv2d_bn_act(C.byref(d), None) == -2
    assert b"2 GiB" in lib.vidc_last_error()


@pytest.mark.parametrize("M,N,K", [(4800, 768, 6912), (80, 3072, 27648), (300, 256, 2304), (19200, 64, 576), (4800, 64, 128)])
def test_conv_plan_is_sane(lib, M, N, K):
    d = L.ConvDesc()
    d.B, d.Ho, d.Wo, d.Cout, d.Cin, d.KH, d.KW, d.groups = 1, 1, M, N, K, 1, 1, 1
    assert lib.vidc_conv2d_plan(C.byref(d)) == 0
    assert 1 <= d.tile < L.TILE_COUNT and d.splitk >= 1
    import re
    bm, bn = [int(v) for v in re.match(r"(\d+)x(\d+)", L.TILE_NAMES[d.tile]).groups()]
    wgs = -(-M // bm) * -(-N // bn) * d.splitk
    assert wgs >= min(64, (M // 32) * (N // 64))        # the planner must not leave most of the 256 CUs idle
    if d.splitk > 1:
        assert lib.vidc_conv2d_workspace_bytes(C.byref(d)) == (L.SPLITK_COUNTERS + d.splitk * M * N) * 4


def test_rng_draws_follow_reference_order():
    """draw_normal_hypotheses / draw_enrichment consume np.random exactly like main.py:43 and :292."""
    from vi_depth_completion_amd import plane, synthetic as S
    ids = S.plane_id_map(240, 320)
    np.random.seed(3)
    slots, hyp, _ = plane.draw_normal_hypotheses([ids])
    np.random.seed(3)
    flat = ids.reshape(-1)
    exp = []
    for cls in (1, 2):
        pix = np.flatnonzero(flat == cls)
        exp.append(pix[np.random.permutation(np.r_[0:len(pix)])[0:300]])
    assert slots.tolist() == [[0, 1, 0, 300], [0, 2, 300, 300]]
    assert np.array_equal(hyp, np.concatenate(exp))
    np.random.seed(4)
    sub, offs = plane.draw_enrichment([1000, 0, 50], 200)
    np.random.seed(4)
    e0 = np.unique(np.random.randint(0, 1000, size=200))
    e2 = np.unique(np.random.randint(0, 50, size=50))
    assert offs.tolist() == [0, len(e0), len(e0), len(e0) + len(e2)]
    assert np.array_equal(sub, np.concatenate([e0, e2]))
    # a plane with > 300 sparse points (known from a first device pass): plane_offset_ransac's permutation (main.py:78) is drawn
    # right after that plane's normal hypotheses and before the next plane's
    np.random.seed(3)
    slots_d, hyp_d, dense = plane.draw_normal_hypotheses([ids], dense={0: 500})
    np.random.seed(3)
    e1 = np.random.permutation(np.r_[0:int((flat == 1).sum())])[0:300]
    ed = np.random.permutation(np.r_[0:500])[0:300]
    e2 = np.random.permutation(np.r_[0:int((flat == 2).sum())])[0:300]
    assert np.array_equal(dense[0], ed) and list(dense) == [0]
    assert np.array_equal(hyp_d, np.concatenate([np.flatnonzero(flat == 1)[e1], np.flatnonzero(flat == 2)[e2]]))
    # background-only map: no slots (main.py:135-137)
    s2, h2, _ = plane.draw_normal_hypotheses([np.zeros((240, 320), np.uint8)])
    assert s2.shape == (0, 4) and h2.size == 0


import contextlib


@contextlib.contextmanager
def _direct_convs():
    """Record with one direct-form launch per conv: no Winograd triples (tests/test_winograd.py::test_recorded_program_has_winograd_triples
    covers that recording)."""
    old = {k: os.environ.get(k) for k in ("VIDC_WINOGRAD",)}
    os.environ["VIDC_WINOGRAD"] = "0"
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def recorded_programs(lib):
    """Both networks recorded on CPU in dry-run mode (no HIP calls): exercises all host-side program logic."""
    import torch
    os.environ["VIDC_PRECISION"] = "mixed"
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval()
    dc = ModifiedFPN().eval()
    with _direct_convs():
        return sn.build_program(1, torch.device("cpu"), dry_run=True), dc.build_program(1, 240, 320, torch.device("cpu"), dry_run=True)


@pytest.fixture(scope="module")
def recorded_frame_program(lib):
    """The software-pipelined tick program (surface-normal net of frame i+1 + depth-completion net of frame i, four
    pyramids per grouped launch), recorded on CPU in dry-run mode."""
    import torch
    os.environ["VIDC_PRECISION"] = "mixed"
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    from vi_depth_completion_amd.pipeline import build_frame_program
    sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval()
    dc = ModifiedFPN().eval()
    with _direct_convs():
        return build_frame_program(sn, dc, 1, 240, 320, torch.device("cpu"), dry_run=True)


def test_opt_in_warp_fusion_rewrites_the_program(lib, monkeypatch):
    """VIDC_FUSE_WARP=1 (off by default: DESIGN 4.4): the surface-normal stem gathers its input through the forward warp, the warp launch
    goes; the three depth-completion stems are untouched."""
    import torch
    monkeypatch.setenv("VIDC_FUSE_WARP", "1")
    monkeypatch.setenv("VIDC_WINOGRAD", "0")
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    from vi_depth_completion_amd.pipeline import build_frame_program
    fp = build_frame_program(SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).eval(), ModifiedFPN().eval(), 1, 240, 320, torch.device("cpu"), dry_run=True)
    kinds = [k for k, _, _, _ in fp.ops]
    assert kinds.count("warp_fwd") == 0 and fp.n_fused_warps == 1 and kinds.count("warp_params") == 1
    stems = [kw for k, _, _, kw in fp.ops if k == "stem"]
    assert [kw.get("warp") is not None for kw in stems] == [True, False, False, False] and stems[0]["key"].startswith("sn/")
    i = kinds.index("stem")
    g, wf = fp.c_ops[i].u.g, stems[0]["warp"]      # engine.finalize -> program.hip launch_op: the record pointer, the principal point, the convention
    assert g.p[4] == fp.storage[wf["p"].buf].data_ptr() + 4 * wf["p"].ch_off and g.p[0] == fp.storage[wf["x"].buf].data_ptr() + 4 * wf["x"].ch_off
    assert (g.f[0], g.f[1], g.i[8]) == (np.float32(wf["intr"].cx), np.float32(wf["intr"].cy), int(wf["ac"]))


def test_frame_program_structure(recorded_frame_program, recorded_programs):
    """Same convolutions as the two separate programs, but the four pyramids share their launches (4 groups), and the
    program is cut into two segments at the surface-normal output."""
    fp = recorded_frame_program
    sn, dc = recorded_programs
    kinds = [k for k, _, _, _ in fp.ops]
    assert kinds.count("stem") == 4 and kinds.count("head") == 2 and kinds.count("maxpool") == 1
    assert kinds.count("warp_fwd") == 1 and fp.n_fused_warps == 0 and kinds.count("warp_inv") == 1 and kinds.count("upsample") == 6
    # 105 grouped pyramid launches + 12 surface-normal decoder/head launches (18 convs) + 12 depth-completion ones
    assert kinds.count("conv") == 105 + 12 + 12
    pyr = [kw for k, _, _, kw in fp.ops if k == "conv" and len(kw["keys"]) == 4]
    assert len(pyr) == 105 and all(kw["keys"][0].startswith("sn/resnet_pyramids.") and kw["keys"][3].startswith("dc/resnet_depth.") for kw in pyr)
    assert fp.flops == sn.flops + dc.flops and fp.ref_flops == sn.ref_flops + dc.ref_flops
    segs = fp.segments()
    assert len(segs) == 2 and segs[0][1] == segs[1][0]
    # segment 0 ends with the inverse warp (the normals the plane block needs); segment 1 holds only depth-completion decoder ops
    assert fp.ops[segs[0][1] - 1][0] == "warp_inv"
    for k, _, _, kw in fp.ops[segs[1][0]:]:
        if k == "conv":
            assert kw["keys"][0].startswith("dc/feature")
    # the decoders read channel slices of the 4-group pyramid levels: group 0 -> surface normal, groups 1..3 -> depth completion
    first_sn = next(kw for k, _, _, kw in fp.ops if k == "conv" and kw["keys"][0] == "sn/feature1_upsamping.0")
    first_dc = next(kw for k, _, _, kw in fp.ops if k == "conv" and kw["keys"][0] == "dc/feature1_upsamping.0")
    assert first_sn["geom"][1] == 256 and first_dc["geom"][1] == 768
    assert first_sn["x"].ch_off == 0 and first_dc["x"].ch_off == 256 and first_sn["x"].ld == first_dc["x"].ld == 1024


def test_joint_weight_store_addresses_both_modules(recorded_frame_program):
    ws = recorded_frame_program.ws
    assert tuple(ws.raw("sn/resnet_pyramids.conv1.conv1_1.weight").shape) == (64, 3, 3, 3)
    assert tuple(ws.raw("dc/resnet_depth.conv1.conv1_1.weight").shape) == (64, 1, 3, 3)
    assert tuple(ws.raw("dc/feature_concat.2.weight").shape) == (1, 192, 1, 1)


def test_program_recording_matches_reference_op_counts(recorded_programs):
    """125 + 337 convs, 293.88 GFLOP per 320x240 frame (46.29 SN + 247.58 DC)."""
    sn, dc = recorded_programs
    kinds = lambda p: [k for k, _, _, _ in p.ops]
    # SN: 1 stem + 105 pyramid launches + 17 decoder convs in 11 launches (same-shape convs of different branches are grouped) + 1 head
    # 3x3 as conv ops, + 1 head 1x1 kernel = 125 convs
    nconv = lambda p: sum(len(kw["keys"]) for k, _, _, kw in p.ops if k == "conv")
    assert kinds(sn).count("stem") == 1 and kinds(sn).count("conv") == 105 + 11 + 1 and nconv(sn) == 123 and kinds(sn).count("head") == 1
    assert kinds(sn).count("upsample") == 3 and kinds(sn).count("maxpool") == 1
    # DC: the 3 pyramids run grouped: 3 stems + 105 grouped launches (= 315 convs) + 17 + 1 + head = 337 convs
    assert kinds(dc).count("stem") == 3 and kinds(dc).count("conv") == 105 + 11 + 1 and nconv(dc) == 315 + 18 and kinds(dc).count("head") == 1
    stem = lambda cin: 2 * 120 * 160 * 64 * cin * 9
    head = lambda cin, co, h, w: 2 * h * w * cin * co
    sn_flops = sn.ref_flops + stem(3) + head(64, 3, 60, 80)
    dc_flops = dc.ref_flops + 2 * stem(3) + stem(1) + head(192, 1, 60, 80)
    assert abs(sn_flops / 1e9 - 46.29) < 0.01, sn_flops
    assert abs(dc_flops / 1e9 - 247.58) < 0.01, dc_flops
    # executed: the six 1x1 convs per network that follow an upsample in the reference run BEFORE it here (4x fewer MACs each)
    assert sn.flops < sn.ref_flops and dc.flops < dc.ref_flops
    # five of them 4x (30x40->60x80, 15x20->30x40), one 3.75x (8x10 -> 15x20), each 2.8312 GFLOP in the reference formulation
    assert abs((dc.ref_flops - dc.flops) / 1e9 - 2.8312 * (5 * 0.75 + (1 - 80 / 300.0))) < 0.01


def test_mixed_precision_program_structure(recorded_programs, recorded_frame_program):
    """bf16x3 convs read a split image that is produced by exactly one preceding split op of the right tensor, and no
    fp32-mode conv reads a split image."""
    for prog in tuple(recorded_programs) + (recorded_frame_program,):
        split_out = {}
        n_bf = 0
        for kind, reads, writes, kw in prog.ops:
            if kind == "split":
                split_out[writes[0]] = kw["x"].buf
            elif kind == "conv":
                if kw["precision"] == L.PREC_BF16X3:
                    n_bf += 1
                    assert kw["x"].buf in split_out, "bf16x3 conv without a split input"
                    assert kw["geom"][1] % 32 == 0
                else:
                    assert kw["x"].buf not in split_out
                if kw.get("split_out") is not None:               # split fused into this conv's epilogue
                    split_out[kw["split_out"].buf] = kw["y"].buf
                    assert kw["flags"] & L.SPLIT_OUT
            elif kind in ("stem", "maxpool", "upsample") and kw.get("split_out") is not None:   # ... or into a glue kernel
                split_out[kw["split_out"].buf] = kw["y"].buf
        assert n_bf >= 3 and len(split_out) <= n_bf
        assert prog.n_fused_splits >= 1
        # every split is written by its producer: no standalone split launch is left in these programs
        assert sum(1 for kind, _, _, _ in prog.ops if kind == "split") == 0
    # the 54-GFLOP-class layer (dc.feature1_upsamping.0) must be on the fast path
    dc = recorded_programs[1]
    big = [kw for kind, _, _, kw in dc.ops if kind == "conv" and kw["keys"][0] == "feature1_upsamping.0"][0]
    assert big["precision"] == L.PREC_BF16X3


def test_program_buffer_reuse_is_safe(recorded_programs, recorded_frame_program):
    """No op may read and write the same storage, and pinned inputs/outputs never alias anything."""
    for prog in tuple(recorded_programs) + (recorded_frame_program,):
        st = prog.storage
        for kind, reads, writes, kw in prog.ops:
            accum = kind in ("conv", "upsample") and kw.get("flags", 0) & (32 if kind == "conv" else 2)
            for w in writes:
                for r in reads:
                    if r == w and accum:
                        continue
                    assert st[r].data_ptr() != st[w].data_ptr(), (kind, r, w)
        pinned = [st[b].data_ptr() for b in prog.pinned]
        assert len(set(pinned)) == len(pinned)
        others = {st[b].data_ptr() for b in range(len(st)) if b not in prog.pinned and st[b] is not None}
        assert not (set(pinned) & others)
        assert prog.bytes_allocated < 600e6


def test_native_permutation_replays_numpy_legacy_stream():
    """vidc_host_mt19937_permutation_prefix (csrc/host_rng.hip) against numpy itself: RandomState.permutation(n)[0:300] draw by draw,
    the generator state afterwards word by word, and the next draw -- for sizes on both sides of the 624-word refill of the MT19937
    state and of the 300-hypothesis cut, from a fresh and from a used generator; np.random (the module) works the same way."""
    from vi_depth_completion_amd import plane
    for seed in (0, 5, 11):
        for n in (0, 1, 2, 3, 17, 299, 300, 301, 623, 624, 625, 1000, 4097, 38400):
            a, b = np.random.RandomState(seed), np.random.RandomState(seed)
            a.random_sample(seed * 7), b.random_sample(seed * 7)
            want = a.permutation(np.r_[0:n])[0:min(300, n)]
            st = plane._LegacyStream.open(b)
            got = st.permutation_prefix(n, 300)
            st.commit()
            assert got.dtype == np.int32 and np.array_equal(want, got), (seed, n)
            sa, sb = a.get_state(), b.get_state()
            assert sa[2] == sb[2] and np.array_equal(sa[1], sb[1]), (seed, n)
            assert a.randint(0, 1 << 30) == b.randint(0, 1 << 30)
    saved = np.random.get_state()
    try:
        np.random.seed(3)
        want = np.random.permutation(np.r_[0:700])[0:300]
        np.random.seed(3)
        st = plane._LegacyStream.open(np.random)
        assert np.array_equal(st.permutation_prefix(700, 300), want)
    finally:
        np.random.set_state(saved)
    assert plane._LegacyStream.open(np.random.default_rng(0)) is None       # not the legacy generator: draw_normal_hypotheses falls back
    lib = L.lib()
    assert lib.vidc_host_mt19937_permutation_prefix(None, None, 4, 2, None, None) != 0

