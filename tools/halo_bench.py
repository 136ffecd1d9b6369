"""3x3x3 stride-1 layers on the halo kernel: lockstep (halo_stagger 0) vs staggered waves (1): time, TF-equivalent, and
bit-identity of the two (SGC_HALO_VARIANTS="1" times the default only; tools/halo_knob_ab.py alternates the values)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):      # diagnostic builds (tools/diag): timing only
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
layers = [  # name, Cin, Cout, grid
    ("256->256 @40x40x16", 256, 256, (40, 40, 16)),
    ("256->128 @40x40x16", 256, 128, (40, 40, 16)),
    ("128->28  @40x40x16", 128, 28, (40, 40, 16)),
    ("512->512 @20x20x8", 512, 512, (20, 20, 8)),
    ("512->128 @20x20x8", 512, 128, (20, 20, 8)),
    ("1024->1024 @10x10x4", 1024, 1024, (10, 10, 4)),
    ("128->128 @80x80x32", 128, 128, (80, 80, 32)),
]
variants = [int(v) for v in os.environ.get("SGC_HALO_VARIANTS", "0,1").split(",")]
for name, Cin, Cout, g in layers:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda")
    wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
    sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    res = {}
    for bd in variants:
        ops.lib.call("sgc_set_tuning", b"halo_stagger", bd)
        f = lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
        for _ in range(3):
            y, og = f()
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 20
        for _ in range(n):
            y, og = f()
        torch.cuda.synchronize(); res[bd] = ((time.perf_counter() - t) / n, y.clone())
    fl = 2 * Cin * Cout * V * 27
    line = f"{name:24s}"
    for bd in variants:
        line += f" | bd{bd} {res[bd][0]*1e6:7.1f} us {fl/res[bd][0]/1e12:6.1f} TF-eq"
    if len(variants) > 1:
        line += f" | identical {all(torch.equal(res[variants[0]][1], res[v][1]) for v in variants[1:])}"
    print(line, flush=True)
