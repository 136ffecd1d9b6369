"""GPU parity of the MFMA implicit-GEMM 3D convolution (through the C ABI) against the naive
oracle loop (itself checked against torch's conv3d / conv_transpose3d on the CPU in
tests/test_oracle_conv.py).  fp32 MFMA products are exact, so the tolerance is summation order."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # Cin, Cout, grid, ksize, stride, transposed, residual, relu
    (32, 128, (10, 10, 4), 3, 1, False, True, True),      # single-pass, fused epilogue
    (64, 128, (8, 6, 4), 3, 2, False, False, True),       # stride 2
    (64, 256, (5, 5, 2), 3, 1, False, True, True),        # few voxels -> split-K + epilogue kernel
    (32, 32, (7, 5, 3), 3, 1, False, False, False),       # narrow tile (Cout <= 32)
    (64, 128, (6, 4, 4), 1, 2, False, False, False),      # 1x1x1 stride-2 downsample
    (64, 128, (5, 4, 3), 2, 2, True, False, True),        # ConvTranspose3d k2 s2
    (256, 256, (12, 12, 8), 3, 1, False, True, True),     # config-2 channel count, multi-block
    (64, 128, (9, 7, 4), 3, 1, False, True, 2),           # decoder epilogue: relu(t) + skip
    (64, 256, (4, 4, 2), 3, 1, False, True, 2),           # same through the split-K epilogue kernel
    # halo-resident kernel (3x3x3 stride 1, >= 2048 voxels): the three brick shapes, partial bricks, split-K
    (64, 128, (16, 16, 16), 3, 1, False, True, True),     # bricks 4x4x16
    (64, 128, (20, 20, 8), 3, 1, False, True, 2),         # bricks 4x8x8, 20 % 8 != 0
    (32, 64, (24, 24, 4), 3, 1, False, False, True),      # bricks 8x8x4
    (64, 192, (17, 13, 16), 3, 1, False, True, True),     # partial bricks in x and y, Cout % 128 != 0
    (32, 256, (40, 40, 16), 3, 1, False, True, True),     # config-2 volume, 200 workgroups, no split-K
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_cl_matches_oracle(case, oracle_ops, gpu_ops):
    Cin, Cout, grid, k, s, tr, use_res, relu = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, Cin, generator=g)
    taps = 8 if tr else k ** 3
    wt = torch.randn(taps, Cout, Cin, generator=g) * (1.0 / (taps * Cin) ** 0.5)
    sc = torch.rand(Cout, generator=g) + 0.5
    sh = torch.randn(Cout, generator=g)
    y_c, og = oracle_ops.conv3d_cl(x, wt, grid, k, s, tr, sc, sh, None, False)
    res = torch.randn(y_c.shape, generator=g) if use_res else None
    y_c, og = oracle_ops.conv3d_cl(x, wt, grid, k, s, tr, sc, sh, res, relu)
    y_g, og_g = gpu_ops.conv3d_cl(x.cuda(), wt.cuda(), grid, k, s, tr, sc.cuda(), sh.cuda(),
                                  res.cuda() if use_res else None, relu)
    assert og == og_g
    err = (y_g.cpu() - y_c).abs().max().item()
    assert err < 2e-5 * max(1.0, y_c.abs().max().item()), err


@pytest.mark.parametrize("case", CASES)
def test_conv3d_bf16x3_is_fp32_faithful(case, oracle_ops, gpu_ops):
    """3-way bf16 split on the bf16 MFMA vs the fp32 oracle: bounded at 1e-4 of the tensor scale
    (observed ~1e-5; north-star bar 1e-3)."""
    Cin, Cout, grid, k, s, tr, use_res, relu = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, Cin, generator=g)
    taps = 8 if tr else k ** 3
    wt = torch.randn(taps, Cout, Cin, generator=g) * (1.0 / (taps * Cin) ** 0.5)
    sc = torch.rand(Cout, generator=g) + 0.5
    sh = torch.randn(Cout, generator=g)
    y_c, og = oracle_ops.conv3d_cl(x, wt, grid, k, s, tr, sc, sh, None, False)
    res = torch.randn(y_c.shape, generator=g) if use_res else None
    y_c, og = oracle_ops.conv3d_cl(x, wt, grid, k, s, tr, sc, sh, res, relu)
    w_hi, w_lo = gpu_ops.split_bf16(wt.cuda())
    y_g, og_g = gpu_ops.conv3d_cl_bf16x3(x.cuda(), w_hi, w_lo, grid, k, s, tr, sc.cuda(), sh.cuda(),
                                         res.cuda() if use_res else None, relu)
    assert og == og_g
    err = (y_g.cpu() - y_c).abs().max().item()
    assert err < 1e-4 * max(1.0, y_c.abs().max().item()), err
    # and the oracle's own bf16x3 entry point (hi + lo recombined) agrees with its fp32 one
    y_c2, _ = oracle_ops.conv3d_cl_bf16x3(x, w_hi.cpu(), w_lo.cpu(), grid, k, s, tr, sc, sh, res, relu)
    assert (y_c2 - y_c).abs().max().item() < 1e-4 * max(1.0, y_c.abs().max().item())
