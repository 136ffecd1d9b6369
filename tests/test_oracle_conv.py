"""The oracle's naive channels-last convolution == torch conv3d / conv_transpose3d (CPU)."""
import pytest
import torch
import torch.nn.functional as F


def cl(x):
    return x[0].permute(1, 2, 3, 0).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("Cin,Cout,g,k,s", [(32, 8, (6, 5, 4), 3, 1), (32, 16, (6, 6, 4), 3, 2), (64, 8, (5, 4, 3), 1, 2)])
def test_conv_matches_torch(Cin, Cout, g, k, s, oracle_ops):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, Cin, *g, generator=gen)
    w = torch.randn(Cout, Cin, k, k, k, generator=gen) * 0.1
    sc, sh = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    ref = F.conv3d(x, w, None, s, k // 2)
    res = torch.randn(ref.shape, generator=gen)
    want = F.relu(ref * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1) + res)
    wt = w.permute(2, 3, 4, 0, 1).reshape(k ** 3, Cout, Cin).contiguous()
    y, og = oracle_ops.conv3d_cl(cl(x), wt, g, k, s, False, sc, sh, cl(res), True)
    assert og == tuple(ref.shape[2:])
    assert (y - cl(want)).abs().max() < 2e-5


def test_transposed_conv_matches_torch(oracle_ops):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(1, 32, 3, 4, 2, generator=gen)
    w = torch.randn(32, 8, 2, 2, 2, generator=gen) * 0.1
    ref = F.conv_transpose3d(x, w, None, 2)
    wt = w.permute(2, 3, 4, 1, 0).reshape(8, 8, 32).contiguous()
    y, og = oracle_ops.conv3d_cl(cl(x), wt, (3, 4, 2), 2, 2, True)
    assert og == (6, 8, 4) and (y - cl(ref)).abs().max() < 1e-5


def test_oracle_upsample_backward_is_the_adjoint_torch_autograd_computes(oracle_ops):
    """sgc_upsample2x_backward restates torch's upsample_trilinear3d backward (align_corners=False, x2): pinned
    by autograd of F.interpolate itself, odd sizes and size-1 axes included."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(4)
    for shape in ((3, 4, 5, 3), (2, 1, 6, 2), (1, 7, 1, 1), (4, 10, 10, 4)):
        x = torch.randn(1, *shape, generator=g, requires_grad=True)
        up = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)
        go = torch.randn(up.shape, generator=g)
        want, = torch.autograd.grad(up, x, go)
        got = oracle_ops.upsample2x_backward(go.contiguous())
        assert got.shape == want.shape and (got - want).abs().max() < 1e-5, shape


def test_oracle_weight_gradient_is_what_torch_autograd_returns(oracle_ops):
    """sgc_conv3d_wgrad_bf16x3 (oracle): nn.Conv3d.weight.grad permuted to [tap][Cout][Cin], strides 1 and 2, 1x1x1, and
    the ConvTranspose3d(2, 2) weight gradient through the k2 s2 geometry with the two tensors exchanged."""
    g = torch.Generator().manual_seed(0)
    for cin, cout, grid, k, s in [(8, 12, (5, 4, 3), 3, 1), (8, 4, (6, 4, 4), 3, 2), (4, 8, (4, 4, 2), 1, 2)]:
        V = grid[0] * grid[1] * grid[2]
        x = torch.randn(V, cin, generator=g)
        w = torch.randn(cout, cin, k, k, k, generator=g).double().requires_grad_(True)
        y = F.conv3d(x.double().view(*grid, cin).permute(3, 0, 1, 2).unsqueeze(0), w, None, s, k // 2)
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(gy)
        dw = oracle_ops.conv3d_wgrad_bf16x3(x, gy[0].permute(1, 2, 3, 0).reshape(-1, cout).float().contiguous(), grid, k, s)
        ref = w.grad.permute(2, 3, 4, 0, 1).reshape(k ** 3, cout, cin)
        assert (dw - ref).abs().max() < 1e-5 * ref.abs().max()
    grid, cin, cout = (3, 4, 2), 8, 4
    x = torch.randn(24, cin, generator=g)
    w = torch.randn(cin, cout, 2, 2, 2, generator=g).double().requires_grad_(True)
    xr = x.double().view(*grid, cin).permute(3, 0, 1, 2).unsqueeze(0).requires_grad_(True)
    y = F.conv_transpose3d(xr, w, None, 2)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    dy = gy[0].permute(1, 2, 3, 0).reshape(-1, cout).float().contiguous()
    dwk = oracle_ops.conv3d_wgrad_bf16x3(dy, x, (6, 8, 4), 2, 2)
    assert (dwk.permute(1, 2, 0).reshape(cin, cout, 2, 2, 2) - w.grad).abs().max() < 1e-5 * w.grad.abs().max()
    wt = w.detach().float().permute(2, 3, 4, 0, 1).reshape(8, cin, cout).contiguous()
    dx, og = oracle_ops.conv3d_cl(dy, wt, (6, 8, 4), 2, 2, False, None, None, None, False)
    assert og == grid and (dx - xr.grad[0].permute(1, 2, 3, 0).reshape(24, cin)).abs().max() < 1e-4


def test_oracle_batch_norm_rows_is_torch_batch_norm(oracle_ops):
    """The oracle's sgc_bn_rows_forward / _backward (double-precision loops, the checker of csrc/batch_norm.hip) against
    F.batch_norm in float64: output, batch statistics, the running-statistics update (momentum, unbiased variance) and the three
    gradients."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    for rows, C in ((37, 8), (600, 64), (5, 4)):
        x = torch.randn(rows, C, generator=g) * 2 + 0.5
        w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
        rm_o, rv_o = rm.clone(), rv.clone()
        y, mean, invstd = oracle_ops.bn_rows_forward(x, w, b, rm_o, rv_o, momentum=0.3, eps=1e-5)
        xd = x.double().requires_grad_(True)
        wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
        rm_r, rv_r = rm.double(), rv.double()
        y_ref = F.batch_norm(xd, rm_r, rv_r, wd, bd, True, 0.3, 1e-5)
        assert torch.allclose(y.double(), y_ref, atol=1e-5)
        assert torch.allclose(mean.double(), xd.mean(0), atol=1e-6)
        assert torch.allclose(invstd.double(), 1 / (xd.var(0, unbiased=False) + 1e-5).sqrt(), rtol=1e-5)
        assert torch.allclose(rm_o.double(), rm_r, atol=1e-6) and torch.allclose(rv_o.double(), rv_r, rtol=1e-5)
        dy = torch.randn(rows, C, generator=g)
        dx, dw, db = oracle_ops.bn_rows_backward(x, dy, mean, invstd, w)
        y_ref.backward(dy.double())
        assert torch.allclose(dx.double(), xd.grad, atol=2e-5)
        assert torch.allclose(dw.double(), wd.grad, atol=2e-4) and torch.allclose(db.double(), bd.grad, atol=2e-4)


def test_oracle_block_diagonal_linear_is_the_per_head_matmul(oracle_ops):
    """Oracle twin of sgc_linear_rows_blockdiag_bf16x3: group g's K inputs times its own [Nh, K] weight -- the V rows of
    nn.MultiheadAttention.in_proj_weight applied head by head (TU/deformable_cross_attention.py:826-833) -- against float64 torch,
    and against nn.MultiheadAttention's own in-projection of V on a feature that is the same for every head."""
    G, K, Nh, rows = 8, 128, 16, 50
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, G * K, generator=g)
    w = torch.randn(G, Nh, K, generator=g) * 0.1
    b = torch.randn(G * Nh, generator=g) * 0.1
    hi, lo = oracle_ops.split_bf16(w)
    y = oracle_ops.linear_rows_blockdiag(x, hi, lo, b)
    want = torch.cat([x[:, h * K:(h + 1) * K].double() @ w[h].double().t() for h in range(G)], 1) + b.double()
    assert float((y.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    mha = torch.nn.MultiheadAttention(K, G)
    wv, bv = mha.in_proj_weight.detach()[2 * K:], mha.in_proj_bias.detach()[2 * K:]
    feat = torch.randn(rows, K, generator=g)
    hi, lo = oracle_ops.split_bf16(wv.view(G, Nh, K).contiguous())
    y = oracle_ops.linear_rows_blockdiag(feat.repeat(1, G).contiguous(), hi, lo, bv)
    want = torch.nn.functional.linear(feat.double(), wv.double(), bv.double())
    assert float((y.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    # the device-side count leaves the rows past it untouched
    cnt = torch.tensor([20], dtype=torch.int32)
    part = oracle_ops.linear_rows_blockdiag(feat.repeat(1, G).contiguous(), hi, lo, bv, count=cnt)
    assert torch.equal(part[:20], y[:20])
