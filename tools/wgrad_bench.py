"""Weight-gradient kernel on the neck's layer shapes: 4 vs 8 waves per tile (wgrad_waves), time, TF-equivalent, bit-identity."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
layers = [("256->256 @40x40x16 k3", 256, 256, (40, 40, 16), 3, 1), ("256->128 @40x40x16 k3", 256, 128, (40, 40, 16), 3, 1),
          ("256->512 s2 k3", 256, 512, (40, 40, 16), 3, 2), ("512->512 @20x20x8 k3", 512, 512, (20, 20, 8), 3, 1),
          ("1024->1024 @10x10x4 k3", 1024, 1024, (10, 10, 4), 3, 1), ("Linear 204800 x 256->256", 256, 256, (204800, 1, 1), 1, 1),
          ("Linear 77000 x 256->512", 256, 512, (77000, 1, 1), 1, 1)]
for name, cin, cout, grid, k, s in layers:
    V = grid[0] * grid[1] * grid[2]
    og = tuple((d + 2 * (k // 2) - k) // s + 1 for d in grid)
    x = torch.randn(V, cin, device="cuda"); dy = torch.randn(og[0] * og[1] * og[2], cout, device="cuda")
    res = {}
    for wv in (4, 8):
        ops.lib.call("sgc_set_tuning", b"wgrad_waves", wv)
        for _ in range(2):
            dw = ops.conv3d_wgrad_bf16x3(x, dy, grid, k, s)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            dw = ops.conv3d_wgrad_bf16x3(x, dy, grid, k, s)
        torch.cuda.synchronize(); res[wv] = ((time.perf_counter() - t) / 10, dw)
    fl = 2.0 * cin * cout * og[0] * og[1] * og[2] * k ** 3
    print(f"{name:28s} 4 waves {res[4][0]*1e6:8.1f} us {fl/res[4][0]/1e12:6.1f} TF-eq | 8 waves {res[8][0]*1e6:8.1f} us {fl/res[8][0]/1e12:6.1f} TF-eq | identical {torch.equal(res[4][1], res[8][1])}", flush=True)
