"""The four-branch FPN-style decoder shared by both networks (networks/surface_normal.py:73-145 and, three
times wider, networks/depth_completion.py:75-147): parameter containers with the reference's Sequential
indices, and the walk that turns them into fused engine ops.
"""
import os

import torch.nn as nn

# per pyramid level: (kernel, cout multiple of 128*m) for each conv, 'u' = upsample to the next finer level
_BRANCH_PLAN = {
    1: [(3, 2), (1, 1)],
    2: [(1, 2), (3, 2), "u", (1, 1)],
    3: [(1, 4), (3, 4), "u", (1, 2), (3, 2), "u", (1, 1)],
    4: [(1, 8), (3, 8), "u", (1, 4), (3, 4), "u", (1, 2), (3, 2), "u", (1, 1)],
}
COMMUTE_UPSAMPLE = os.environ.get("VIDC_COMMUTE_UPSAMPLE", "1") == "1"
_LEVEL_SIZES_240 = {1: (60, 80), 2: (30, 40), 3: (15, 20)}   # the reference's literals, used only as nn metadata


def build_branch(level, m):
    """nn.Sequential whose indices/shapes equal `feature{level}_upsamping` of the reference (m = width multiplier)."""
    mods = []
    cin = 256 * m * (2 ** (level - 1))
    target = level
    for step in _BRANCH_PLAN[level]:
        if step == "u":
            target -= 1
            mods.append(nn.UpsamplingBilinear2d(size=_LEVEL_SIZES_240[target]))
            continue
        k, mult = step
        cout = 128 * m * mult
        mods += [nn.Conv2d(cin, cout, k, 1, k // 2), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
        cin = cout
    return nn.Sequential(*mods)


def emit_branch(prog, seq, prefix, x, level_sizes, zsum):
    """Walks one branch.  Conv2d+BatchNorm2d+ReLU triples become one fused conv; the last conv of the branch writes
    (branch 1) or accumulates (branches 2-4) into `zsum`, which implements z1+z2+z3+z4 without add kernels.

    UpsamplingBilinear2d -> Conv2d(1x1) -> BatchNorm2d -> ReLU (every upsample of the reference is followed by exactly that,
    surface_normal.py:88-91 ...) is executed as conv1x1+BN at the LOW resolution -> upsample -> ReLU: a 1x1 conv and a
    per-channel affine commute with bilinear interpolation (its weights sum to 1), so the result is the same up to fp32
    rounding while the conv does 4x fewer MACs and the upsample moves half the channels."""
    mods = list(seq)
    n_conv = sum(isinstance(mm, nn.Conv2d) for mm in mods)
    seen = 0
    t = x
    i = 0
    while i < len(mods):
        mm = mods[i]
        if isinstance(mm, nn.Conv2d):
            assert isinstance(mods[i + 1], nn.BatchNorm2d) and isinstance(mods[i + 2], nn.ReLU)
            seen += 1
            last = seen == n_conv
            pad = mm.padding[0] if isinstance(mm.padding, tuple) else mm.padding
            if last and zsum is not None:
                t = prog.conv(t, "%s%d" % (prefix, i), bn="%s%d" % (prefix, i + 1), relu=True, padding=pad, out=zsum,
                              accumulate=True)
            else:
                t = prog.conv(t, "%s%d" % (prefix, i), bn="%s%d" % (prefix, i + 1), relu=True, padding=pad)
            i += 3
        elif isinstance(mm, nn.UpsamplingBilinear2d):
            # target size = the next finer pyramid level of *this* input resolution (the reference hard-codes 240x320)
            level = [lv for lv, sz in level_sizes.items() if sz == (t.H, t.W)][0] - 1
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if COMMUTE_UPSAMPLE and isinstance(nxt, nn.Conv2d) and nxt.kernel_size == (1, 1):
                assert isinstance(mods[i + 2], nn.BatchNorm2d) and isinstance(mods[i + 3], nn.ReLU)
                seen += 1
                last = seen == n_conv
                Hn, Wn = level_sizes[level]
                t = prog.conv(t, "%s%d" % (prefix, i + 1), bn="%s%d" % (prefix, i + 2), relu=False,        # low resolution
                              ref_flops_scale=(Hn * Wn) / float(t.H * t.W))
                t = prog.upsample(t, level_sizes[level], relu=True, into=zsum if (last and zsum is not None) else None)
                i += 4
            else:
                t = prog.upsample(t, level_sizes[level])
                i += 1
        else:
            raise TypeError(type(mm))
    return t


GROUP_BRANCHES = os.environ.get("VIDC_GROUP_DECODER", "1") == "1"


def emit_decoder_grouped(prog, module, levels, key_prefix=""):
    """The same four branches, walked level by level instead of branch by branch: convs of different branches that have the same
    shape at the same pyramid level run as ONE grouped launch on channel slices of a shared buffer --
      level 3: 3x3 (f3.3, f4.10) and the low-resolution 1x1 (f3.7, f4.14) as 2 groups,
      level 2: 3x3 (f2.3, f3.10, f4.17) and the low-resolution 1x1 (f2.7, f3.14, f4.21) as 3 groups --
    17 -> 11 conv launches and 6 -> 3 upsample launches per decoder, same arithmetic per output element (grouping only changes
    which launch computes it; z1+z2+z3+z4 is still accumulated in that order).  Needs the upsample/1x1 commute (emit_branch)."""
    from ..engine import K, T
    flat = [T(t.buf, t.B, t.H, t.W, t.C * t.G, 1, t.ld, t.ch_off) for t in levels]
    x1, x2, x3, x4 = flat
    size = {i + 1: (t.H, t.W) for i, t in enumerate(flat)}
    f = lambda b, i: "%sfeature%d_upsamping.%d" % (key_prefix, b, i)
    seqs = {b: list(getattr(module, "feature%d_upsamping" % b)) for b in (1, 2, 3, 4)}
    for b, idxs in ((2, (0, 3, 7)), (3, (0, 3, 7, 10, 14)), (4, (0, 3, 7, 10, 14, 17, 21))):      # the Sequential layout this relies on
        assert all(isinstance(seqs[b][i], nn.Conv2d) for i in idxs) and sum(isinstance(mm, nn.Conv2d) for mm in seqs[b]) == len(idxs)

    def cbr(x, b, i, relu=True, **kw):                      # Conv2d + BatchNorm2d (+ ReLU) of branch b at Sequential index i
        k = seqs[b][i].kernel_size[0]
        return prog.conv(x, f(b, i), bn=f(b, i + 1), relu=relu, padding=k // 2, **kw)

    def grouped(x, pairs, relu=True, **kw):                 # the same layer shape of several branches in one launch
        k = seqs[pairs[0][0]][pairs[0][1]].kernel_size[0]
        return prog.conv(x, K([f(b, i) for b, i in pairs]), bn=K([f(b, i + 1) for b, i in pairs]), relu=relu, padding=k // 2, **kw)

    sl = lambda t, g, n=1: T(t.buf, t.B, t.H, t.W, t.C, n, t.ld, t.ch_off + g * t.C)      # groups g .. g+n-1 of a grouped tensor
    up_scale = lambda lo, hi: (size[hi][0] * size[hi][1]) / float(size[lo][0] * size[lo][1])
    # level 4 (branch 4 only)
    t = cbr(cbr(x4, 4, 0), 4, 3)
    t = cbr(t, 4, 7, relu=False, ref_flops_scale=up_scale(4, 3))                  # low resolution; ReLU after the upsample
    # level 3: [f3 | f4]
    c3 = seqs[3][3].in_channels
    u3 = prog.nhwc(size[3][0], size[3][1], c3, 2)
    cbr(x3, 3, 0, out=sl(u3, 0))
    prog.upsample(t, size[3], relu=True, out=sl(u3, 1))
    g3 = grouped(u3, ((3, 3), (4, 10)))
    g3 = grouped(g3, ((3, 7), (4, 14)), relu=False, ref_flops_scale=up_scale(3, 2))
    # level 2: [f2 | f3 | f4]
    c2 = seqs[2][3].in_channels
    u2 = prog.nhwc(size[2][0], size[2][1], c2, 3)
    cbr(x2, 2, 0, out=sl(u2, 0))
    prog.upsample(g3, size[2], relu=True, out=sl(u2, 1, 2))
    g2 = grouped(u2, ((2, 3), (3, 10), (4, 17)))
    g2 = grouped(g2, ((2, 7), (3, 14), (4, 21)), relu=False, ref_flops_scale=up_scale(2, 1))
    # level 1: z1, then += z2, z3, z4 in the reference's order
    zsum = cbr(cbr(x1, 1, 0), 1, 3)
    prog.upsample(g2, size[1], relu=True, into=zsum, sum_groups=True)        # (((z1 + z2) + z3) + z4), one launch
    return zsum


def emit_decoder(prog, module, levels, key_prefix=""):
    """levels: [x1..x4] program tensors (possibly grouped = channel-concatenated).  Returns z1+z2+z3+z4.
    key_prefix: prepended to the parameter names (a Program over an engine.JointWeightStore addresses "name/param")."""
    if COMMUTE_UPSAMPLE and GROUP_BRANCHES:
        return emit_decoder_grouped(prog, module, levels, key_prefix)
    from ..engine import T
    flat = [T(t.buf, t.B, t.H, t.W, t.C * t.G, 1, t.ld, t.ch_off) for t in levels]   # concat view: groups -> channels
    sizes = {i + 1: (t.H, t.W) for i, t in enumerate(flat)}
    zsum = None
    for b in (1, 2, 3, 4):
        seq = getattr(module, "feature%d_upsamping" % b)
        z = emit_branch(prog, seq, "%sfeature%d_upsamping." % (key_prefix, b), flat[b - 1], sizes, zsum)
        if zsum is None:
            zsum = z
    return zsum
