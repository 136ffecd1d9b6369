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
    (128, 28, (20, 20, 8), 3, 1, False, False, False),    # the head's fused centerness/reg/cls conv: 28 of 128 tile columns live
    # whole-grid bricks of the coarsest scales (>= 512 output channels): 400 voxels on 512 MFMA rows, 2 x 288 on 384
    (64, 512, (10, 10, 4), 3, 1, False, True, True),
    (96, 576, (12, 12, 4), 3, 1, False, True, 2),
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


@pytest.mark.parametrize("cin,cout,grid,stride", [(1024, 1024, (10, 10, 4), 1), (256, 512, (40, 40, 16), 2),
                                                    (512, 128, (20, 20, 8), 1)])
def test_split_k_layers_are_bitwise_reproducible(cin, cout, grid, stride, gpu_ops):
    """Layers with few output voxels split their reduction over workgroups; with the workspace TensorOps hands
    over, the partial tiles are summed in a fixed order: identical bits on every launch, also while another
    stream keeps the chip busy.  (Without a workspace the library falls back to float atomics: same values to
    rounding.)"""
    g = torch.Generator().manual_seed(17)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g).cuda()
    wt = (torch.randn(27, cout, cin, generator=g) * 0.02).cuda()
    w_hi, w_lo = gpu_ops.split_bf16(wt)
    n_ws = int(gpu_ops.lib._dll.sgc_conv3d_workspace_floats(*grid, cin, cout, 3, stride, 0, 1))
    assert n_ws > 0                                   # the layer IS split
    ref, og = gpu_ops.conv3d_cl_bf16x3(x, w_hi, w_lo, grid, 3, stride, False)
    ref = ref.clone()
    side = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(5):
        with torch.cuda.stream(side):
            busy @ busy
        y, _ = gpu_ops.conv3d_cl_bf16x3(x, w_hi, w_lo, grid, 3, stride, False)
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
    # the atomics fallback (no workspace) agrees to rounding
    y2 = torch.empty_like(ref)
    gpu_ops._call("sgc_conv3d_cl_bf16x3", x, w_hi, w_lo, None, None, None, y2, *grid, cin, cout, 3, stride, 0, 0, None, 0)
    torch.cuda.synchronize()
    assert float((y2 - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("grid,cin,cout,relu,res", [((40, 40, 16), 64, 128, 1, False), ((20, 12, 8), 32, 32, 2, True),
                                                     ((16, 16, 4), 64, 28, 0, False)])
def test_output_masked_conv_is_bit_identical_on_live_rows(grid, cin, cout, relu, res, gpu_ops):
    """sgc_conv3d_cl_bf16x3_masked: rows with mask 1 equal the dense launch bit for bit, rows with mask 0 are finite;
    clustered masks (dead bricks and dead 64-voxel tiles exist), an all-zero mask and an all-one mask."""
    g = torch.Generator().manual_seed(sum(grid) + cin)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g).cuda()
    w = (torch.randn(27, cout, cin, generator=g) * 0.05)
    hi, lo = gpu_ops.split_bf16(w)
    hi, lo = hi.cuda(), lo.cuda()
    scale, shift = torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda()
    residual = torch.randn(V, cout, generator=g).cuda() if res else None
    dense, _ = gpu_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu)
    blob = torch.zeros(grid)
    blob[: grid[0] // 3, grid[1] // 4: grid[1] // 2, :] = 1           # a slab: whole bricks dead elsewhere
    blob[-3:, -2:, -1:] = 1                                              # and a small corner cluster
    for mask in (blob, torch.zeros(grid), torch.ones(grid), (torch.rand(grid, generator=g) < 0.02).float()):
        m = mask.reshape(-1).to(torch.uint8).cuda()
        got, _ = gpu_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu, out_mask=m)
        assert torch.isfinite(got).all()
        live = m.bool()
        assert torch.equal(got[live], dense[live])


@pytest.mark.parametrize("grid,cin,cout,relu,res", [((40, 40, 16), 256, 256, 1, True), ((12, 9, 19), 64, 160, 2, False),
                                                     ((20, 20, 8), 128, 128, 0, False), ((10, 12, 4), 96, 28, 1, True),
                                                     ((17, 6, 8), 512, 256, 0, False), ((8, 8, 4), 32, 128, 0, False)])
def test_halo_staggered_schedule_is_bit_identical_to_the_lockstep_form(grid, cin, cout, relu, res, gpu_ops):
    """The halo kernel's staggered schedule (waves 4-7 half a tap behind waves 0-3, weight tiles published at the start of an
    interval; `halo_stagger` 1, the default) accumulates every output in the same (tap, k-half, product) order as the lockstep
    form (0): every brick shape, ragged grids, split-K slices, column counts that are not a multiple of 128, one channel slice
    (the laggers' trailing half-step is the whole second half of the last tap), an output mask; repeated launches, because a
    missing barrier between the laggers' fragment reads and the next weight tile would show as a rare mismatch."""
    g = torch.Generator().manual_seed(sum(grid) + cin + cout)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g).cuda()
    w = (torch.randn(27, cout, cin, generator=g) * 0.05)
    hi, lo = gpu_ops.split_bf16(w)
    hi, lo = hi.cuda(), lo.cuda()
    scale, shift = torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda()
    residual = torch.randn(V, cout, generator=g).cuda() if res else None
    mask = (torch.rand(grid, generator=g) < 0.3).reshape(-1).to(torch.uint8).cuda()
    try:
        gpu_ops.lib.call("sgc_set_tuning", b"halo_min_m", 1)
        gpu_ops.lib.call("sgc_set_tuning", b"halo_stagger", 0)
        lock, _ = gpu_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu)
        lock_m, _ = gpu_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu, out_mask=mask)
        gpu_ops.lib.call("sgc_set_tuning", b"halo_stagger", 1)
        for _ in range(8):
            stg, _ = gpu_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu)
            assert torch.equal(stg, lock)
        stg_m, _ = gpu_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu, out_mask=mask)
        assert torch.equal(stg_m[mask.bool()], lock_m[mask.bool()])
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"halo_stagger", 1)
        gpu_ops.lib.call("sgc_set_tuning", b"halo_min_m", 2048)


def test_k2_s2_convolution_matches_oracle(oracle_ops, gpu_ops):
    """ksize 2 / stride 2 / no padding (the input gradient of ConvTranspose3d(2, 2)), fp32 and bf16x3 kernels."""
    g = torch.Generator().manual_seed(5)
    grid, Cin, Cout = (8, 6, 4), 64, 96
    x = torch.randn(grid[0] * grid[1] * grid[2], Cin, generator=g)
    wt = torch.randn(8, Cout, Cin, generator=g) * 0.05
    y_c, og = oracle_ops.conv3d_cl(x, wt, grid, 2, 2, False, None, None, None, False)
    assert og == (4, 3, 2)
    y_g, og_g = gpu_ops.conv3d_cl(x.cuda(), wt.cuda(), grid, 2, 2, False, None, None, None, False)
    assert og_g == og and float((y_g.cpu() - y_c).abs().max()) < 2e-5 * float(y_c.abs().max())
    hi, lo = gpu_ops.split_bf16(wt.cuda())
    y_b, _ = gpu_ops.conv3d_cl_bf16x3(x.cuda(), hi, lo, grid, 2, 2)
    assert float((y_b.cpu() - y_c).abs().max()) < 1e-4 * float(y_c.abs().max())


WGRAD_CASES = [  # Cin, Cout, grid, ksize, stride
    (64, 128, (10, 9, 4), 3, 1),
    (32, 28, (7, 5, 6), 3, 1),             # head-like: Cout % 128 != 0
    (256, 256, (24, 20, 8), 3, 1),         # several tiles, voxel range split over workgroups (workspace reduce)
    (64, 128, (8, 6, 4), 3, 2),
    (64, 132, (6, 4, 4), 1, 2),
    (96, 64, (8, 6, 4), 2, 2),             # ConvTranspose3d(2, 2) with the roles exchanged
    (36, 4, (5, 5, 3), 1, 1),
    (64, 96, (9, 10, 5), 3, 1),            # halo form with one 32-channel dy tile per workgroup, ragged bricks in x, y and z
    (32, 64, (17, 8, 4), 3, 1),            # halo form, a single (co, ci) tile, three bricks in one workgroup
]


def test_wgrad_halo_forms_agree_with_the_tile_kernel(oracle_ops, gpu_ops):
    """The halo form of the weight gradient (tuning key wgrad_halo: 1 = double-buffered bricks of 4 x 8 x 4, the default; 2 = single-buffered bricks of 8 x 8 x 4) and the
    per-tap tile kernel (0) against the oracle on one layer whose brick range is split over workgroups."""
    Cin, Cout, grid = 64, 128, (24, 17, 9)
    g = torch.Generator().manual_seed(11)
    V = grid[0] * grid[1] * grid[2]
    x, dy = torch.randn(V, Cin, generator=g), torch.randn(V, Cout, generator=g)
    ref = oracle_ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
    try:
        for mode in (0, 1, 2):
            gpu_ops.lib.call("sgc_set_tuning", b"wgrad_halo", mode)
            got = gpu_ops.conv3d_wgrad_bf16x3(x.cuda(), dy.cuda(), grid, 3, 1).cpu()
            assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), mode
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"wgrad_halo", 1)


def test_wgrad_halo_forms_on_random_shapes(gpu_ops):
    """Seeded random layer shapes (ragged bricks in every direction, 1 - 40 voxels per axis, Cin / Cout multiples of 32 up to
    192) through both halo forms of the 3x3x3 weight gradient against the per-tap tile kernel, which the cases below pin to the
    oracle; repeated launches are bit-identical (tools/wgrad_fuzz.py runs more of them)."""
    import random
    rnd = random.Random(5)
    try:
        for case in range(14):
            Cin, Cout = 32 * rnd.randint(1, 6), 32 * rnd.randint(1, 6)
            grid = (rnd.randint(1, 40), rnd.randint(1, 24), rnd.randint(1, 18))
            g = torch.Generator().manual_seed(case)
            V = grid[0] * grid[1] * grid[2]
            x, dy = torch.randn(V, Cin, generator=g).cuda(), torch.randn(V, Cout, generator=g).cuda()
            gpu_ops.lib.call("sgc_set_tuning", b"wgrad_halo", 0)
            ref = gpu_ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
            for mode in (1, 2):
                gpu_ops.lib.call("sgc_set_tuning", b"wgrad_halo", mode)
                got = gpu_ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
                assert torch.equal(got, gpu_ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)), (Cin, Cout, grid, mode)
                assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (Cin, Cout, grid, mode)
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"wgrad_halo", 1)


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv3d_wgrad_matches_oracle(case, oracle_ops, gpu_ops):
    """sgc_conv3d_wgrad_bf16x3 against the double-accumulating oracle loop: the 3-way bf16 split is fp32-faithful (bound
    1e-4 of the tensor scale as for the forward kernel); two launches are bit-identical (ordered split reduction)."""
    Cin, Cout, grid, k, s = case
    g = torch.Generator().manual_seed(sum(grid) + Cin + Cout)
    V = grid[0] * grid[1] * grid[2]
    pad = 0 if k == 2 else k // 2
    og = tuple((d + 2 * pad - k) // s + 1 for d in grid)
    x = torch.randn(V, Cin, generator=g)
    dy = torch.randn(og[0] * og[1] * og[2], Cout, generator=g)
    ref = oracle_ops.conv3d_wgrad_bf16x3(x, dy, grid, k, s)
    got = gpu_ops.conv3d_wgrad_bf16x3(x.cuda(), dy.cuda(), grid, k, s)
    again = gpu_ops.conv3d_wgrad_bf16x3(x.cuda(), dy.cuda(), grid, k, s)
    assert got.shape == ref.shape == (k ** 3, Cout, Cin)
    assert torch.equal(got, again)
    assert float((got.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout,grid,k,s", [(64, 96, (10, 8, 6), 3, 1), (32, 64, (8, 6, 4), 3, 2), (64, 128, (6, 6, 4), 1, 2),
                                                (128, 25, (9, 7, 4), 3, 1)])
def test_channels_last_conv_function_matches_torch_autograd(cin, cout, grid, k, s, gpu_ops):
    """ChannelsLastConv3dFunction (forward, input gradient, weight gradient on the HIP kernels) against torch's conv3d
    autograd in float64 on the CPU."""
    from sgcdet_amd.functions import ChannelsLastConv3dFunction
    g = torch.Generator().manual_seed(cin + cout)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) * 0.05
    xr = x.double().view(*grid, cin).permute(3, 0, 1, 2).unsqueeze(0).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = torch.nn.functional.conv3d(xr, wr, None, s, k // 2)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = ChannelsLastConv3dFunction.apply(xg, wg, grid, k, s)
    y_ref = yr[0].permute(1, 2, 3, 0).reshape(-1, cout)
    assert float((y.detach().cpu() - y_ref).abs().max()) < 1e-4 * float(y_ref.abs().max())
    y.backward(gy[0].permute(1, 2, 3, 0).reshape(-1, cout).float().cuda())
    dx_ref = xr.grad[0].permute(1, 2, 3, 0).reshape(V, cin)
    assert float((xg.grad.cpu() - dx_ref).abs().max()) < 1e-4 * float(dx_ref.abs().max())
    assert float((wg.grad.cpu() - wr.grad).abs().max()) < 1e-4 * float(wr.grad.abs().max())


def test_channels_last_conv_transpose_function_matches_torch_autograd(gpu_ops):
    from sgcdet_amd.functions import ChannelsLastConvTranspose3dFunction
    g = torch.Generator().manual_seed(11)
    grid, cin, cout = (5, 4, 3), 64, 32
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(cin, cout, 2, 2, 2, generator=g) * 0.05
    xr = x.double().view(*grid, cin).permute(3, 0, 1, 2).unsqueeze(0).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = torch.nn.functional.conv_transpose3d(xr, wr, None, 2)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = ChannelsLastConvTranspose3dFunction.apply(xg, wg, grid)
    y_ref = yr[0].permute(1, 2, 3, 0).reshape(-1, cout)
    assert float((y.detach().cpu() - y_ref).abs().max()) < 1e-4 * float(y_ref.abs().max())
    y.backward(gy[0].permute(1, 2, 3, 0).reshape(-1, cout).float().cuda())
    dx_ref = xr.grad[0].permute(1, 2, 3, 0).reshape(V, cin)
    assert float((xg.grad.cpu() - dx_ref).abs().max()) < 1e-4 * float(dx_ref.abs().max())
    assert float((wg.grad.cpu() - wr.grad).abs().max()) < 1e-4 * float(wr.grad.abs().max())


@pytest.mark.parametrize("N,H,W,cin,cout,k,relu,res", [(3, 15, 20, 64, 96, 3, 1, True), (2, 8, 10, 256, 32, 1, 0, False),
                                                        (4, 30, 40, 32, 256, 3, 0, False), (1, 7, 5, 96, 28, 3, 2, True),
                                                        (5, 37, 45, 64, 28, 3, 1, True), (2, 64, 80, 256, 256, 3, 2, True),
                                                        (7, 16, 20, 96, 160, 3, 0, False)])
def test_conv2d_nhwc_matches_oracle_and_torch(N, H, W, cin, cout, k, relu, res, oracle_ops, gpu_ops):
    """sgc_conv2d_nhwc_bf16x3 (row f-1: the FPN's convolutions on channels-last image rows): against the oracle loop and
    against F.conv2d on the NCHW view; images must not leak into each other (the taps are confined to one image).  The 3 x 3
    cases with >= 2048 pixels run the 2-D form of the halo kernel (16 x 16-pixel bricks: ragged rows / columns, 28 and 160
    output channels, several images per launch), the others the tile kernel; the two agree to fp32 summation noise."""
    g = torch.Generator().manual_seed(N * H + cin + cout)
    x = torch.randn(N * H * W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.05
    wt = w.permute(2, 3, 0, 1).reshape(k * k, cout, cin).contiguous()
    hi, lo = gpu_ops.split_bf16(wt)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    r = torch.randn(N * H * W, cout, generator=g) if res else None
    ref = oracle_ops.conv2d_nhwc_bf16x3(x, hi, lo, (N, H, W), k, sc, sh, r, relu)
    got = gpu_ops.conv2d_nhwc_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), (N, H, W), k, sc.cuda(), sh.cuda(),
                                     r.cuda() if res else None, relu).cpu()
    assert float((got - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    if relu == 0 and not res:
        y = torch.nn.functional.conv2d(x.view(N, H, W, cin).permute(0, 3, 1, 2), (hi.float() + lo.float()).view(k, k, cout, cin).permute(2, 3, 0, 1),
                                       None, 1, k // 2)
        y = y.permute(0, 2, 3, 1).reshape(-1, cout) * sc + sh
        assert float((got - y).abs().max()) < 1e-4 * max(1.0, float(y.abs().max()))
    if k == 3 and N * H * W >= 2048:
        try:
            gpu_ops.lib.call("sgc_set_tuning", b"halo_2d", 0)
            tile = gpu_ops.conv2d_nhwc_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), (N, H, W), k, sc.cuda(), sh.cuda(),
                                              r.cuda() if res else None, relu).cpu()
        finally:
            gpu_ops.lib.call("sgc_set_tuning", b"halo_2d", 1)
        assert float((got - tile).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        again = gpu_ops.conv2d_nhwc_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), (N, H, W), k, sc.cuda(), sh.cuda(),
                                           r.cuda() if res else None, relu).cpu()
        assert torch.equal(again, got)                              # one split, no atomics: the same bits every run


@pytest.mark.parametrize("rows,cin,cout,bias", [(5000, 256, 256, True), (1237, 128, 96, True), (777, 64, 33, False)])
def test_linear_rows_function_matches_torch_autograd(rows, cin, cout, bias, gpu_ops):
    """LinearRowsFunction (forward / input gradient on sgc_linear_rows_bf16x3, weight gradient on the ksize-1 wgrad kernel)
    against float64 autograd of F.linear; output widths that need padding to 4 / 32 columns included."""
    from sgcdet_amd.functions import LinearRowsFunction
    g = torch.Generator().manual_seed(rows + cout)
    x = torch.randn(rows, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) * 0.05
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    y = LinearRowsFunction.apply(xg, wg, bg)
    assert y.shape == (rows, cout)
    assert float((y.detach().cpu() - yr.detach()).abs().max()) < 1e-4 * float(yr.abs().max())
    y.backward(gy.float().cuda())
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 1e-4 * float(xr.grad.abs().max())
    assert float((wg.grad.cpu() - wr.grad).abs().max()) < 1e-4 * float(wr.grad.abs().max())
    if bias:
        assert float((bg.grad.cpu() - br.grad).abs().max()) < 1e-4 * float(br.grad.abs().max())


def test_plain_bf16_mode_against_the_oracle_and_its_error_bound(oracle_ops, gpu_ops):
    """sgc_set_conv_products(1): the opt-in reduced-precision mode of BASELINE.json configs #2 / #5 (reference: the fp16 twin
    of the operator, TU/multi_scale_3ddeformable_attn_function.py:353-428).  Same kernels with ONE bf16 product per
    multiply-add.  (a) against the oracle in the same mode (operands rounded to bf16, fp32 sums): 1e-5 of the scale;
    (b) against the fp32 oracle: within 2^-7 of the output scale (measured 1.5e-3 .. 4e-3 on these layers: two operands
    at 2^-9 relative each, random signs) -- the bound this mode is quoted with."""
    g = torch.Generator().manual_seed(11)
    cases = [("halo 3x3x3", 64, 64, (8, 8, 16), 3, 1), ("strided 3x3x3", 64, 96, (8, 8, 8), 3, 2), ("row GEMM", 256, 256, (300, 1, 1), 1, 1)]
    try:
        for name, cin, cout, grid, k, s in cases:
            V = grid[0] * grid[1] * grid[2]
            x = torch.randn(V, cin, generator=g)
            w = torch.randn(k ** 3, cout, cin, generator=g) * (1.0 / (cin * k ** 3) ** 0.5)
            sh = torch.randn(cout, generator=g) * 0.1
            hi, lo = gpu_ops.split_bf16(w)
            ref32, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, k, s, False, None, sh, None, 1)
            gpu_ops.lib.call("sgc_set_conv_products", 1)
            oracle_ops.lib.call("sgc_set_conv_products", 1)
            y, _ = gpu_ops.conv3d_cl_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), grid, k, s, False, None, sh.cuda(), None, 1)
            ref16, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, k, s, False, None, sh, None, 1)
            gpu_ops.lib.call("sgc_set_conv_products", 3)
            oracle_ops.lib.call("sgc_set_conv_products", 3)
            y3, _ = gpu_ops.conv3d_cl_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), grid, k, s, False, None, sh.cuda(), None, 1)
            scale = max(1.0, float(ref32.abs().max()))
            assert float((y.cpu() - ref16).abs().max()) <= 1e-5 * scale, name
            err16 = float((y.cpu() - ref32).abs().max()) / scale
            err3 = float((y3.cpu() - ref32).abs().max()) / scale
            assert err3 <= 1e-4 and 1e-4 < err16 <= 2.0 ** -7, (name, err16, err3)
    finally:
        gpu_ops.lib.call("sgc_set_conv_products", 3)
        oracle_ops.lib.call("sgc_set_conv_products", 3)


def test_plain_fp16_mode_against_the_oracle_and_its_error_bound(oracle_ops, gpu_ops):
    """sgc_set_conv_products(2): the opt-in fp16 mode (BASELINE.json config #5; reference: the fp16 twin of the operator,
    TU/multi_scale_3ddeformable_attn_function.py:353-428) -- ONE v_mfma_f32_32x32x16_f16 product per multiply-add, operands
    rounded to IEEE half and saturated at +-65504.  (a) against the oracle in the same mode (integer-arithmetic half
    conversion, fp32 sums): 1e-5 of the scale, on the halo, tile and row-GEMM kernels and the 2-D form; (b) against the fp32
    oracle: within 2^-10 of the output scale and at least 4x closer than the bf16 mode on the same layer (11 vs 8 significant
    bits); (c) an activation far beyond the half range saturates instead of producing infinities."""
    g = torch.Generator().manual_seed(12)
    cases = [("halo 3x3x3", 64, 64, (8, 8, 16), 3, 1), ("strided 3x3x3", 64, 96, (8, 8, 8), 3, 2), ("row GEMM", 256, 256, (300, 1, 1), 1, 1),
             ("narrow halo", 32, 28, (8, 16, 8), 3, 1)]
    try:
        for name, cin, cout, grid, k, s in cases:
            V = grid[0] * grid[1] * grid[2]
            x = torch.randn(V, cin, generator=g)
            w = torch.randn(k ** 3, cout, cin, generator=g) * (1.0 / (cin * k ** 3) ** 0.5)
            sh = torch.randn(cout, generator=g) * 0.1
            hi, lo = gpu_ops.split_bf16(w)
            h16, z16 = gpu_ops.split_f16(w)
            assert torch.equal(h16.view(torch.float16).float(), w.half().float())
            ref32, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, k, s, False, None, sh, None, 1)
            for ops in (gpu_ops, oracle_ops):
                ops.lib.call("sgc_set_conv_products", 1)
            y_b, _ = gpu_ops.conv3d_cl_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), grid, k, s, False, None, sh.cuda(), None, 1)
            for ops in (gpu_ops, oracle_ops):
                ops.lib.call("sgc_set_conv_products", 2)
            y, _ = gpu_ops.conv3d_cl_bf16x3(x.cuda(), h16.cuda(), z16.cuda(), grid, k, s, False, None, sh.cuda(), None, 1)
            ref16, _ = oracle_ops.conv3d_cl_bf16x3(x, h16, z16, grid, k, s, False, None, sh, None, 1)
            for ops in (gpu_ops, oracle_ops):
                ops.lib.call("sgc_set_conv_products", 3)
            scale = max(1.0, float(ref32.abs().max()))
            assert float((y.cpu() - ref16).abs().max()) <= 1e-5 * scale, name
            err16 = float((y.cpu() - ref32).abs().max()) / scale
            errb = float((y_b.cpu() - ref32).abs().max()) / scale
            assert 1e-6 < err16 <= 2.0 ** -10 and err16 * 4 <= errb, (name, err16, errb)
        # saturation: 1e6 is not representable in half; the product of the saturated operand stays finite
        x = torch.full((256, 32), 1.0e6)
        w = torch.full((1, 32, 32), 2.0 ** -10)
        h16, z16 = gpu_ops.split_f16(w)
        gpu_ops.lib.call("sgc_set_conv_products", 2)
        y, _ = gpu_ops.conv3d_cl_bf16x3(x.cuda(), h16.cuda(), z16.cuda(), (256, 1, 1), 1, 1, False, None, None, None, 0)
        assert torch.isfinite(y).all() and torch.allclose(y.cpu(), torch.full((256, 32), 65504.0 * 32 * 2.0 ** -10))
    finally:
        gpu_ops.lib.call("sgc_set_conv_products", 3)
        oracle_ops.lib.call("sgc_set_conv_products", 3)


@pytest.mark.parametrize("shape", [(96, 64, 3, 3, 3), (70, 33, 3, 3, 3), (64, 128, 2, 2, 2), (130, 256), (28, 128, 3, 3, 3)])
def test_weight_pack_and_unpack_equal_their_torch_formulation(shape, oracle_ops, gpu_ops):
    """sgc_pack_conv_weight / sgc_unpack_conv_wgrad (the training step's layout passes, one launch each) == the permute /
    flip / pad / split chain of torch ops they replace, bit for bit, on the GPU and in the oracle."""
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(*shape, generator=g)
    A, B = shape[0], shape[1]
    T = w.numel() // (A * B)
    w3 = w.reshape(A, B, T)
    for transpose, flip, pr, pc in ((False, False, 4, 1), (True, True, 1, 32), (True, False, 1, 1), (False, False, 1, 32)):
        src = w3.flip(2) if flip else w3
        ref = src.permute(2, 1, 0) if transpose else src.permute(2, 0, 1)              # [T, R, C]
        R, C = -(-ref.shape[1] // pr) * pr, -(-ref.shape[2] // pc) * pc
        full = torch.zeros(T, R, C)
        full[:, :ref.shape[1], :ref.shape[2]] = ref
        hi_ref, lo_ref = gpu_ops.split_bf16(full)
        for ops, dev in ((gpu_ops, "cuda"), (oracle_ops, "cpu")):
            hi, lo = ops.pack_conv_weight(w.to(dev), transpose=transpose, flip=flip, pad_rows=pr, pad_cols=pc)
            assert hi.shape == (T, R, C) and torch.equal(hi.cpu(), hi_ref) and torch.equal(lo.cpu(), lo_ref), (transpose, flip, dev)
            back = ops.unpack_conv_wgrad(full.to(dev), w.shape, transpose=transpose, flip=flip)
            assert torch.equal(back.cpu(), w), (transpose, flip, dev)


def test_batched_weight_pack_is_the_single_pack(oracle_ops, gpu_ops):
    """sgc_pack_conv_weight_batch (every parameter of a training step in one launch) writes, per item, the bits of
    sgc_pack_conv_weight -- ragged shapes, all four forms, 27 / 8 / 1 taps in one list; the padding of the planes stays as
    allocated (zero); the oracle's batch twin agrees."""
    g = torch.Generator().manual_seed(11)
    shapes = [(96, 64, 27), (70, 33, 27), (64, 128, 8), (130, 256, 1), (28, 128, 27), (256, 256, 27), (8, 8, 1), (33, 7, 8)]
    forms = ((False, False, 4, 1), (True, True, 1, 32), (True, False, 1, 1), (False, False, 1, 32))
    for ops, dev in ((gpu_ops, "cuda"), (oracle_ops, "cpu")):
        entries, want = [], []
        for i, shp in enumerate(shapes):
            w = torch.randn(*shp, generator=g).to(dev)
            for transpose, flip, pr, pc in (forms[i % 4], forms[(i + 1) % 4]):
                T, R, C = ops.packed_shape(shp, transpose, pr, pc)
                hi = torch.zeros(T, R, C, dtype=torch.bfloat16, device=dev)
                lo = torch.zeros_like(hi)
                entries.append((w, hi, lo, transpose, flip))
                want.append(ops.pack_conv_weight(w, transpose=transpose, flip=flip, pad_rows=pr, pad_cols=pc))
        plan = ops.pack_conv_weight_plan(entries)
        ops.run_pack_plan(plan)
        for (w, hi, lo, transpose, flip), (whi, wlo) in zip(entries, want):
            assert torch.equal(hi.view(torch.int16), whi.view(torch.int16)) and torch.equal(lo.view(torch.int16), wlo.view(torch.int16)), (tuple(w.shape), transpose, flip, dev)
        for w, *_ in entries:                          # new values, same storage: the plan is reusable
            w.mul_(1.5)
        ops.run_pack_plan(plan)
        w, hi, lo, transpose, flip = entries[3]
        T, R, C = hi.shape
        again = ops.pack_conv_weight(w, transpose=transpose, flip=flip, pad_rows=R if R > 1 else 1, pad_cols=C if C > 1 else 1)
        assert torch.equal(hi.view(torch.int16), again[0].view(torch.int16)) and torch.equal(lo.view(torch.int16), again[1].view(torch.int16))


def test_train_weight_planes_repack_once_per_step_and_never_serve_stale():
    """functions.TrainWeightPlanes: planes are registered on first use, `begin_step` repacks all of them in one launch, an in-place
    update of a parameter between `begin_step` and the use is seen (version counter), per-step views of one parameter share an
    entry, a parameter that moved gets new planes and the entry of its old storage is dropped after two unused steps."""
    from sgcdet_amd import ext
    from sgcdet_amd.functions import TrainWeightPlanes
    ops = ext.ops()
    tp = TrainWeightPlanes()
    ws = [torch.nn.Parameter(torch.randn(64, 32, 3, 3, 3, device="cuda")), torch.nn.Parameter(torch.randn(120, 96, device="cuda"))]
    same = lambda a, b: all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(a, b))   # noqa: E731
    fresh = lambda w, **kw: ops.pack_conv_weight(w.detach(), **kw)                                         # noqa: E731
    forms = [dict(pad_rows=4), dict(transpose=True, flip=True, pad_cols=32)]
    for w in ws:
        for kw in forms:
            assert same(tp.get(w, **kw), fresh(w, **kw))
    assert same(tp.get(ws[1][40:80], **forms[0]), fresh(ws[1][40:80], **forms[0]))       # a view (in_proj_weight[C:2C])
    assert tp.launches == 0 and len(tp.entries) == 5
    with torch.no_grad():
        for w in ws:
            w.add_(1.0)                                # an optimizer step
    tp.begin_step()
    assert tp.launches == 1
    for w in ws:
        for kw in forms:
            assert same(tp.get(w, **kw), fresh(w, **kw))
    assert same(tp.get(ws[1][40:80], **forms[0]), fresh(ws[1][40:80], **forms[0]))       # a NEW view object, the same entry
    assert tp.launches == 1 and len(tp.entries) == 5   # all served from the batch
    with torch.no_grad():
        ws[0].mul_(0.5)                                # changed after begin_step: the version counter catches it
    assert same(tp.get(ws[0], **forms[0]), fresh(ws[0], **forms[0])) and tp.launches == 2
    ws[1].data = ws[1].data.clone()                    # the parameter moved: new planes, the plan is rebuilt
    assert same(tp.get(ws[1], **forms[1]), fresh(ws[1], **forms[1])) and len(tp.entries) == 6
    for _ in range(4):
        tp.begin_step()
        assert same(tp.get(ws[1], **forms[1]), fresh(ws[1], **forms[1]))
    assert len(tp.entries) == 1                        # everything not asked for during two steps is gone
    hi, lo = tp.get(ws[1], **forms[1])
    assert hi.shape[2] == 128 and not hi[:, :, 120:].any()        # the zero padding of the planes survives the batch
    # a per-step temporary (the head's torch.cat of three parameters): marked ephemeral, it is packed where it is
    # used, shared by the step's later uses and NEVER registered -- no new entry, no rebuilt plan, gone at the next begin_step
    plans_before, n_before = tp.plans, len(tp.entries)
    for step in range(3):
        cat = torch.cat([ws[0].detach(), ws[0].detach() * 2.0], 0)
        tp.mark_ephemeral(cat)
        first = tp.get(cat, **forms[0])
        assert same(first, fresh(cat, **forms[0]))
        assert tp.get(cat.detach(), **forms[0])[0] is first[0]       # the alias a Function sees: the same planes, packed once
        assert same(tp.get(cat, **forms[1]), fresh(cat, **forms[1]))
        assert len(tp.entries) == n_before and tp.plans is plans_before
        tp.begin_step()
        assert not tp.ephemeral
    # stand-alone callers never reach begin_step: the table is capped
    tp2 = TrainWeightPlanes()
    tp2.MAX_ENTRIES = 8
    keep = [torch.randn(32, 32, device="cuda") for _ in range(20)]
    for w in keep:
        tp2.get(w)
    assert len(tp2.entries) <= 8


@pytest.mark.parametrize("cin,cout,grid,k,s,tr", [(64, 128, (8, 8, 8), 3, 2, False), (128, 256, (10, 10, 4), 3, 1, False),
                                                  (512, 128, (8, 8, 4), 2, 2, True), (96, 192, (13, 7, 5), 3, 2, False)])
def test_tile_kernel_options_are_bit_identical(cin, cout, grid, k, s, tr, oracle_ops, gpu_ops):
    """The tile implicit GEMM's options -- the XCD deals of its workgroups (`igemm_xcd` 1 / 2), four waves per tile
    (`conv_waves`) -- only move work around: same K order, same split, same bits as the default,
    which itself stays within the bf16x3 bound of the oracle (ragged row / column counts, a stride, a transposed layer)."""
    g = torch.Generator().manual_seed(cin + cout + 7)
    V = grid[0] * grid[1] * grid[2]
    taps = 8 if tr else k ** 3
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(taps, cout, cin, generator=g) * (1.0 / (cin * (1 if tr else taps)) ** 0.5)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    hi, lo = gpu_ops.split_bf16(w)
    y_o, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, k, s, tr, sc, sh, None, 1)
    args = (x.cuda(), hi.cuda(), lo.cuda(), grid, k, s, tr, sc.cuda(), sh.cuda(), None, 1)
    y_d, _ = gpu_ops.conv3d_cl_bf16x3(*args)
    assert float((y_d.cpu() - y_o).abs().max()) <= 1e-4 * max(1.0, float(y_o.abs().max()))
    for key, val, back in ((b"igemm_xcd", 1, 0), (b"igemm_xcd", 2, 0), (b"conv_waves", 4, 8)):
        try:
            gpu_ops.lib.call("sgc_set_tuning", key, val)
            y_v, _ = gpu_ops.conv3d_cl_bf16x3(*args)
        finally:
            gpu_ops.lib.call("sgc_set_tuning", key, back)
        assert torch.equal(y_v, y_d), (key, val)


@pytest.mark.parametrize("cin,cout,grid,k,s,tr,min_steps,res,relu", [
    (160, 128, (8, 6, 4), 3, 2, False, 8, True, 1),      # 135 K steps in 15 splits of 9: boundaries INSIDE taps (5 chunks per tap)
    (256, 96, (5, 4, 3), 2, 2, True, 2, False, 1),       # transposed: the 8 channel chunks of a parity in 4 splits
    (96, 128, (6, 6, 4), 3, 2, False, 1, False, 0),      # one split per K step (81 steps, capped at 32 splits of 3)
    (64, 200, (7, 5, 3), 3, 1, False, 4, True, 2),       # ragged rows / columns, 54 steps in 11 splits of 5 (the last one holds 4)
])
def test_tile_kernel_splits_of_whole_steps_against_the_oracle(cin, cout, grid, k, s, tr, min_steps, res, relu, oracle_ops, gpu_ops):
    """Round 6: the tile kernel splits a reduction at any K step (32 channels of one tap), not only between groups of taps
    (`pick_split_steps`, csrc/conv3d.hip).  A split that starts and ends inside a tap must decode its (tap, chunk) start, walk
    into the next tap and stop short of the last one's end: against the oracle within the bf16x3 bound, bitwise reproducible,
    and within summation order of the tap-group splits it replaces."""
    g = torch.Generator().manual_seed(cin * 3 + cout)
    V = grid[0] * grid[1] * grid[2]
    taps = 8 if tr else k ** 3
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(taps, cout, cin, generator=g) * (1.0 / (cin * (1 if tr else taps)) ** 0.5)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    hi, lo = gpu_ops.split_bf16(w)
    y0, og = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, k, s, tr, sc, sh, None, 0)
    r = torch.randn(y0.shape, generator=g) if res else None
    y_o, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, k, s, tr, sc, sh, r, relu)
    args = (x.cuda(), hi.cuda(), lo.cuda(), grid, k, s, tr, sc.cuda(), sh.cuda(), r.cuda() if res else None, relu)
    try:
        gpu_ops.lib.call("sgc_set_tuning", b"split_min_steps", min_steps)
        n_ws = int(gpu_ops.lib._dll.sgc_conv3d_workspace_floats(*grid, cin, cout, k, s, 1 if tr else 0, 1))
        assert n_ws >= 3 * y_o.numel()                       # the layer IS split, at least three ways
        y_a, og_g = gpu_ops.conv3d_cl_bf16x3(*args)
        y_b, _ = gpu_ops.conv3d_cl_bf16x3(*args)
        gpu_ops.lib.call("sgc_set_tuning", b"split_free", 0)
        y_t, _ = gpu_ops.conv3d_cl_bf16x3(*args)
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"split_free", 1)
        gpu_ops.lib.call("sgc_set_tuning", b"split_min_steps", 8)
    assert tuple(og_g) == tuple(og)
    scale = max(1.0, float(y_o.abs().max()))
    assert float((y_a.cpu() - y_o).abs().max()) <= 1e-4 * scale
    assert torch.equal(y_a, y_b)
    assert float((y_a - y_t).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("grid,cin,cout", [((40, 40, 16), 128, 28), ((20, 20, 8), 128, 28), ((10, 10, 4), 128, 28), ((12, 12, 4), 64, 200),
                                           ((24, 24, 8), 128, 24)])
def test_conv_with_the_head_activation_in_its_epilogue(grid, cin, cout, oracle_ops, gpu_ops):
    """sgc_conv3d_cl_bf16x3_act: columns [1, 7) leave as exp(v * scale) -- ImVoxelHeadV2's exp(scale(reg)) inside the fused
    centerness | reg | cls convolution (dense_heads/imvoxel_head_v2.py:79,103-110) -- on every kernel the head's three scales reach
    (halo bricks with 64-column tiles, the tile kernel with a split reduction and its epilogue kernel), with and without an
    output mask.  Against the oracle twin; the untouched columns and torch's own exp(scale * x) of the plain launch, bit for bit."""
    g = torch.Generator().manual_seed(sum(grid) + cout)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(27, cout, cin, generator=g) * 0.02
    hi, lo = gpu_ops.split_bf16(w)
    shift = torch.randn(cout, generator=g) * 0.1
    s = torch.tensor(0.83)
    want, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, None, shift, None, 0, act=(1, 7, s))
    cu = lambda t: t.cuda()
    plain, _ = gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, None, cu(shift), None, 0)
    got, _ = gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, None, cu(shift), None, 0, act=(1, 7, cu(s)))
    assert (got.cpu() - want).abs().max() < 1e-4 * max(1.0, float(want.abs().max()))
    assert torch.equal(got[:, :1], plain[:, :1]) and torch.equal(got[:, 7:], plain[:, 7:])
    ref = torch.exp(plain[:, 1:7] * cu(s))                                     # what the two torch launches computed
    assert ((got[:, 1:7] - ref).abs() <= 2.4e-7 * ref.abs()).all()             # same product, exp within 2 ulp of torch's
    mask = (torch.rand(V, generator=g) < 0.3).to(torch.uint8).cuda()
    mask[: V // 3] = 0                                                          # dead bricks
    gm, _ = gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, None, cu(shift), None, 0, out_mask=mask, act=(1, 7, cu(s)))
    assert torch.isfinite(gm).all() and torch.equal(gm[mask.bool()], got[mask.bool()])
    with pytest.raises(RuntimeError):
        gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, None, cu(shift), None, 0, act=(5, 5, cu(s)))


@pytest.mark.parametrize("grid,cin,cout,relu,res", [((40, 40, 16), 128, 28, 0, False), ((20, 20, 8), 128, 28, 1, True), ((24, 24, 8), 256, 32, 2, True),
                                                     ((17, 9, 16), 64, 20, 0, False), ((16, 16, 32), 32, 8, 1, False)])
def test_32_column_tiles_are_bit_identical_to_the_64_column_form(grid, cin, cout, relu, res, oracle_ops, gpu_ops):
    """Round 5: layers with <= 32 output channels (the head's fused 28-column convolution) run the halo kernel with 32-column
    tiles and an 8 x 1 wave layout instead of 64 columns of which more than half were padding.  Every output sees the same
    (tap, k-half, product) order: bit-identical to the 64-column form (`halo_narrow` = 64) on every brick shape, ragged grids,
    split channel slices, with residual / relu modes and an output mask; and within tolerance of the oracle."""
    g = torch.Generator().manual_seed(sum(grid) + cin + cout)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(27, cout, cin, generator=g) * 0.05
    hi, lo = gpu_ops.split_bf16(w)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    residual = torch.randn(V, cout, generator=g) if res else None
    cu = lambda t: None if t is None else t.cuda()
    mask = (torch.rand(V, generator=g) < 0.3).to(torch.uint8).cuda()
    mask[: V // 4] = 0
    outs = {}
    try:
        for form in (64, 1):
            gpu_ops.lib.call("sgc_set_tuning", b"halo_narrow", form)
            outs[form] = [gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, cu(scale), cu(shift), cu(residual), relu)[0],
                          gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, cu(scale), cu(shift), cu(residual), relu, out_mask=mask)[0]]
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"halo_narrow", 1)
    assert torch.equal(outs[1][0], outs[64][0])
    live = mask.bool()
    assert torch.isfinite(outs[1][1]).all() and torch.equal(outs[1][1][live], outs[64][0][live])
    want, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu)
    assert (outs[1][0].cpu() - want).abs().max() < 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("grid,cin,cout,relu,res", [((16, 16, 16), 64, 128, 1, True), ((40, 40, 16), 256, 256, 1, True), ((8, 24, 8), 96, 160, 2, True),
                                                     ((16, 8, 32), 32, 72, 0, False), ((20, 20, 8), 512, 512, 1, True), ((13, 11, 8), 64, 96, 0, False)])
def test_winograd_z_convolution_against_the_direct_form(grid, cin, cout, relu, res, oracle_ops, gpu_ops):
    """sgc_conv3d_winograd_z_bf16x3 (F(2,3) along z: 18 of the 27 tap-GEMMs) is the same operator as the direct 3x3x3 kernel:
    against the ORACLE's direct convolution on the original weights and against the GPU's direct kernel, 2e-5 of the tensor scale
    (the bound the round-4 review set); against its own oracle twin 1e-5; borders in z (the zero rows d0 / d3 of the first / last
    pair), residual and both relu modes; repeated launches bit-identical; unsupported shapes refused."""
    g = torch.Generator().manual_seed(sum(grid) + cin)
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(V, cin, generator=g)
    w = torch.randn(27, cout, cin, generator=g) * (1.0 / (27 * cin) ** 0.5)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    residual = torch.randn(V, cout, generator=g) if res else None
    hi, lo = gpu_ops.split_bf16(w)
    ghi, glo = gpu_ops.split_bf16(gpu_ops.winograd_z_weights(w))
    cu = lambda t: None if t is None else t.cuda()
    assert gpu_ops.conv3d_winograd_z_supported(grid, cin, cout)
    got, _ = gpu_ops.conv3d_winograd_z(cu(x), cu(ghi), cu(glo), grid, cu(scale), cu(shift), cu(residual), relu)
    again, _ = gpu_ops.conv3d_winograd_z(cu(x), cu(ghi), cu(glo), grid, cu(scale), cu(shift), cu(residual), relu)
    assert torch.equal(got, again)
    direct_gpu, _ = gpu_ops.conv3d_cl_bf16x3(cu(x), cu(hi), cu(lo), grid, 3, 1, False, cu(scale), cu(shift), cu(residual), relu)
    want, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, grid, 3, 1, False, scale, shift, residual, relu)
    sc = max(1.0, float(want.abs().max()))
    assert (got.cpu() - want).abs().max() < 2e-5 * sc, float((got.cpu() - want).abs().max()) / sc
    assert (got - direct_gpu).abs().max() < 2e-5 * sc
    if V * cin * cout <= 16 * 16 * 16 * 64 * 128:                   # the literal Winograd restatement of the oracle (slow loops)
        twin, _ = oracle_ops.conv3d_winograd_z(x, ghi, glo, grid, scale, shift, residual, relu)
        assert (got.cpu() - twin).abs().max() < 1e-5 * sc
    assert not gpu_ops.conv3d_winograd_z_supported((16, 16, 12), cin, cout) and not gpu_ops.conv3d_winograd_z_supported(grid, cin, 64)
    assert not gpu_ops.conv3d_winograd_z_supported((13, 9, 8), cin, cout)          # fewer than 2048 rows in the stack: the direct kernel


def test_winograd_z_through_the_neck_keeps_parity():
    """conv_plan.WINOGRAD_Z on / off through the whole neck + head on one volume: every head tensor within 1e-4 of its scale of the
    direct form (twelve chained convolutions), and the form under test really ran."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd import ext
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin import conv_plan
    from sgcdet_amd.scene import model_config, workload
    w = workload("cfg2_scannet")
    torch.manual_seed(2)
    det = build_detector(model_config(w)).eval().cuda()
    det.use_graph = det.scene_graph = False
    vol = torch.randn(1, w["embed_dims"], *w["n_voxels_list"][-1], device="cuda")
    outs = {}
    ops = ext.ops()
    try:
        for mode in (False, True):
            conv_plan.set_winograd_z(mode, min_channels=256)       # True (not "auto"): also the layers a lone scene would leave direct
            det.neck_3d.__dict__.pop("_hip_plan", None)
            ops.event_log, ops.event_names = [], {"sgc_conv3d_winograd_z_bf16x3"}
            with torch.no_grad():
                o = det._neck_head_eager(vol)
            torch.cuda.synchronize()
            n_w = len(ops.event_log)
            ops.event_log, ops.event_names = None, None
            assert (n_w == 7) == bool(mode), n_w            # 40 x 40 x 16: three 256 -> 256, one 256 -> 128; 20 x 20 x 8: two 512 -> 512, one 512 -> 128
            outs[mode] = [t.clone() for part in o for t in part]
    finally:
        conv_plan.set_winograd_z("auto", min_channels=256)
        det.neck_3d.__dict__.pop("_hip_plan", None)
    for a, b in zip(outs[True], outs[False]):
        assert (a - b).abs().max() <= 1e-4 * max(1.0, float(b.abs().max()))
