"""The streamed grouped GEMM tile (csrc/wgemm.hip, VIDC_TILE_G96x32_STREAM) behind vidc_conv2d_bn_act: the Winograd-domain products of the small
maps, M[gi] = V[gi] U[gi]^T with at most 96 rows and many groups (the 3x3 layers of networks/surface_normal.py:27-50 / depth_completion.py:75-147
in their F(m x m, 3x3) form).  Against a float64 CPU product; against the general tile (same K split: bit for bit); result bits
independent of the chunking and of which other groups share the launch (what engine.Program.group_variant relies on)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TILE_STREAMS, TILE_GENERAL = (40, 41), 28       # g96x32s, g96x64s3; 32x64k2d2
TILE_STREAM = 40


def _case(M, G, K, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((1, 1, M, G * K), generator=g)
    w = torch.randn((G, N, K), generator=g) * (1.0 / np.sqrt(K))
    return x, w


def _run(x, w, tile, G, chunks=1, relu=False, scale=None, shift=None):
    from vi_depth_completion_amd import ops
    N = w.shape[1]
    s = torch.ones(N) if scale is None else scale
    b = torch.zeros(N) if shift is None else shift
    y = ops.conv2d_bn_act(x.to(DEV), w.to(DEV).reshape(G, N, -1), s.to(DEV), b.to(DEV), 1, 1, relu1=relu, tile=tile, splitk=chunks, groups=G)
    torch.cuda.synchronize()
    return y.cpu()


@pytest.mark.parametrize("tile", TILE_STREAMS)
@pytest.mark.parametrize("M,G,K,N", [(80, 144, 256, 256), (80, 64, 512, 512), (96, 36, 128, 64), (20, 36, 256, 64), (33, 5, 192, 128), (80, 16, 1024, 128)])
def test_streamed_tile_against_float64_and_general_tile(M, G, K, N, tile):
    x, w = _case(M, G, K, N, seed=M + G)
    got = _run(x, w, tile, G)
    ref = torch.einsum("mgk,gnk->mgn", x.view(M, G, K).double(), w.double()).reshape(1, 1, M, G * N)
    err = float((got.double() - ref).abs().max())
    assert err < 2e-4 * float(ref.abs().max()), err
    gen = _run(x, w, TILE_GENERAL, G)
    assert torch.equal(got, gen), "the two tiles split K the same way (units 2 st + kq, slice 0 + slice 1): bits must agree"


@pytest.mark.parametrize("TILE_STREAM", TILE_STREAMS)
def test_bits_do_not_depend_on_chunks_or_on_the_groups_in_the_launch(TILE_STREAM):
    M, G, K, N = 80, 144, 256, 256
    x, w = _case(M, G, K, N, seed=3)
    ref = _run(x, w, TILE_STREAM, G)
    for chunks in (2, 7, 16, 64, 144, 1000):
        assert torch.equal(_run(x, w, TILE_STREAM, G, chunks=chunks), ref), chunks
    # a restriction to a contiguous range of groups = pointer offsets (engine.Program.group_variant): groups 36..107 alone
    xs = x.view(1, 1, M, G, K)[:, :, :, 36:108].reshape(1, 1, M, 72 * K).contiguous()
    part = _run(xs, w[36:108].contiguous(), TILE_STREAM, 72)
    assert torch.equal(part.view(M, 72, N), ref.view(M, G, N)[:, 36:108])


def test_affine_relu_and_refusals():
    from vi_depth_completion_amd import ops
    M, G, K, N = 80, 8, 128, 64
    x, w = _case(M, G, K, N, seed=9)
    s, b = torch.rand(N) + 0.5, torch.randn(N)
    got = _run(x, w, TILE_STREAM, G, relu=True, scale=s, shift=b)
    ref = torch.relu(torch.einsum("mgk,gnk->mgn", x.view(M, G, K).double(), w.double()) * s.double() + b.double()).reshape(1, 1, M, G * N)
    assert float((got.double() - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    with pytest.raises(RuntimeError, match="rows"):           # more than 96 rows
        _run(*_case(128, 4, 128, 64), TILE_STREAM, 4)
    with pytest.raises(RuntimeError, match="Cin"):            # K = 64: a group must span at least two ring stages
        _run(*_case(80, 4, 64, 64), TILE_STREAM, 4)
