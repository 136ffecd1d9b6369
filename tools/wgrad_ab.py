"""Weight-gradient kernels A/B: the halo form (tuning key wgrad_halo = 1: double-buffered bricks of 4 x 8 x 4, the product; 2: single-buffered bricks of 8 x 8 x 4) against the per-tap tile kernel (0) on the 3x3x3
stride-1 layer shapes of the neck, alternated in one process; checks that both agree.  Usage: python tools/wgrad_ab.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext  # noqa: E402

ops = ext.ops()
SHAPES = [(256, 256, (40, 40, 16)), (128, 128, (40, 40, 16)), (256, 512, (20, 20, 8)), (512, 512, (20, 20, 8)),
          (1024, 1024, (10, 10, 4)), (256, 256, (80, 80, 32))]
for Cin, Cout, grid in SHAPES:
    V = grid[0] * grid[1] * grid[2]
    if V * max(Cin, Cout) * 4 > 2 ** 31:
        continue
    g = torch.Generator().manual_seed(1)
    x = torch.randn(V, Cin, generator=g).cuda()
    dy = torch.randn(V, Cout, generator=g).cuda()
    res, t = {}, {0: [], 1: [], 2: []}
    for rep in range(6):
        for mode in (0, 1, 2):
            ops.lib.call("sgc_set_tuning", b"wgrad_halo", mode)
            for _ in range(2):
                res[mode] = ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv3d_wgrad_bf16x3(x, dy, grid, 3, 1)
            e1.record()
            torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / 5 * 1e3)
    ops.lib.call("sgc_set_tuning", b"wgrad_halo", 1)
    fl = 2.0 * V * Cin * Cout * 27 * 3
    m0, m1 = sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]
    err = float((res[0] - res[1]).abs().max() / res[0].abs().max())
    print(json.dumps(dict(Cin=Cin, Cout=Cout, grid=grid, tile_us=round(m0, 1), halo_us=round(m1, 1), halo_single_buffered_us=round(sorted(t[2])[len(t[2]) // 2], 1), ratio=round(m1 / m0, 3),
                          halo_frac_of_mfma_peak=round(fl / (m1 * 1e-6) / 2.5e15, 3), rel_diff=err)), flush=True)
