"""Random layer shapes through the halo forms of the 3x3x3 weight gradient (wgrad_halo 1 and 2) against the per-tap tile kernel
(0, itself pinned to the oracle by tests/test_gpu_conv3d.py): grids with ragged bricks in every direction, 1 - 40 voxels per
axis, Cin / Cout in {32 .. 192}.  Usage: python tools/wgrad_fuzz.py [cases] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext  # noqa: E402

ops = ext.ops()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, halo_cases = 0.0, 0
for case in range(n_cases):
    Cin, Cout = 32 * rnd.randint(1, 6), 32 * rnd.randint(1, 6)
    grid = (rnd.randint(1, 40), rnd.randint(1, 24), rnd.randint(1, 18))
    V = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(case)
    x = torch.randn(V, Cin, generator=g).cuda()
    dy = torch.randn(V, Cout, generator=g).cuda()
    ops.lib.call("sgc_set_tuning", b"wgrad_halo", 0)
    ref = ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
    scale = float(ref.abs().max()) + 1e-30
    ws = {m: int(ops.lib._dll.sgc_conv3d_wgrad_workspace_floats(*grid, Cin, Cout, 3, 1)) for m in (0,)}
    for mode in (1, 2):
        ops.lib.call("sgc_set_tuning", b"wgrad_halo", mode)
        got = ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
        again = ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
        err = float((got - ref).abs().max()) / scale
        worst = max(worst, err)
        assert torch.equal(got, again), ("not deterministic", Cin, Cout, grid, mode)
        assert err < 2e-5, (Cin, Cout, grid, mode, err)
    ops.lib.call("sgc_set_tuning", b"wgrad_halo", 1)
print(f"{n_cases} shapes, both halo forms agree with the tile kernel: max |diff| / max|dW| = {worst:.2e}; launches repeat bit for bit")
