"""CPU restatement of the Winograd F(m x m, 3x3) transforms of csrc/winograd.hip -- TEST INFRASTRUCTURE (only tests/ may import it).

The reference has no Winograd code: its 3x3 layers are `nn.Conv2d(c, c, 3, 1, 1)` (networks/surface_normal.py:75,84,96,...;
networks/depth_completion.py:77,86,98,...) executed by ATen, so the pin of the Winograd path is `F.conv2d` itself (tests/test_winograd.py
compares the composed HIP path with it) and, through the networks, the reference-generated goldens.  This file restates the three
transforms with the matrices of Lavin & Gray (2015) in the kernels' layouts so each kernel can be checked on its own.
"""
import numpy as np
import torch


def matrices(m):
    if m == 2:
        BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    elif m == 4:
        BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
        G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    else:
        raise ValueError(m)
    return (torch.tensor(BT, dtype=torch.float64), torch.tensor(G, dtype=torch.float64), torch.tensor(AT, dtype=torch.float64))


def weight_transform(w, m):
    """(Cout,Cin,3,3) -> (a*a, Cout, Cin) fp32, computed in fp64."""
    _BT, G, _AT = matrices(m)
    u = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G)
    return u.reshape(-1, w.shape[0], w.shape[1]).float()


def input_transform(x_nhwc, cin, m):
    """(B,H,W,G*cin) -> (tiles, G*a*a*cin), rows [gg][pos][cin]; fp64 arithmetic (the kernel's fp32 result differs by rounding only)."""
    BT, _G, _AT = matrices(m)
    a = m + 2
    B, H, W, Cc = x_nhwc.shape
    G = Cc // cin
    th, tw = -(-H // m), -(-W // m)
    xp = torch.zeros((B, th * m + 2, tw * m + 2, Cc), dtype=torch.float64)
    xp[:, 1:H + 1, 1:W + 1] = x_nhwc.double()
    p = xp.unfold(1, a, m).unfold(2, a, m)                       # (B, th, tw, Cc, a, a)
    v = torch.einsum("ij,bytcjk,lk->bytcil", BT, p, BT)          # (B, th, tw, Cc, a, a)
    v = v.reshape(B * th * tw, G, cin, a * a).permute(0, 1, 3, 2)
    return v.reshape(B * th * tw, G * a * a * cin).float()


def output_transform(mm, B, Ho, Wo, cout, m):
    """(tiles, G*a*a*cout) -> (B,Ho,Wo,G*cout) = A^T M A, no epilogue; fp64 arithmetic."""
    _BT, _G, AT = matrices(m)
    a = m + 2
    th, tw = -(-Ho // m), -(-Wo // m)
    G = mm.shape[1] // (a * a * cout)
    t = mm.double().reshape(B, th, tw, G, a, a, cout)
    y = torch.einsum("pi,bytgilc,sl->byptsgc", AT, t, AT)        # (B, th, m, tw, m, G, cout)
    y = y.reshape(B, th * m, tw * m, G * cout)[:, :Ho, :Wo]
    return y.float().contiguous()


def conv3x3(x_nhwc, w_groups, m):
    """The whole Winograd conv on the CPU (fp64 inside): for the tests' sanity check of this restatement against F.conv2d."""
    G = len(w_groups)
    B, H, W, Cc = x_nhwc.shape
    cin, cout = Cc // G, w_groups[0].shape[0]
    a2 = (m + 2) * (m + 2)
    v = input_transform(x_nhwc, cin, m).double().reshape(-1, G, a2, cin)
    u = torch.stack([weight_transform(w, m) for w in w_groups]).double()          # (G, a2, cout, cin)
    mm = torch.einsum("tgpc,gpoc->tgpo", v, u).reshape(v.shape[0], G * a2 * cout).float()
    return output_transform(mm, B, H, W, cout, m)
