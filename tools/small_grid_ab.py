"""The 3x3x3 stride-1 layers of the coarsest scales (10x10x4 of config 2, 12x12x4 of config 3) on whole-grid bricks of the halo
kernel (`halo_small` 1) against the tile kernel (0): alternated rounds, results compared within fp32 summation order."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Cin, Cout, g in [("1024->1024 @10x10x4", 1024, 1024, (10, 10, 4)), ("1024->128 @10x10x4", 1024, 128, (10, 10, 4)),
                           ("1024->1024 @12x12x4", 1024, 1024, (12, 12, 4)), ("1024->128 @12x12x4", 1024, 128, (12, 12, 4))]:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); wt = torch.randn(27, Cout, Cin, device="cuda") * (1.0 / (27 * Cin) ** 0.5)
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.randn(Cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    f = lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    ts = {0: [], 1: []}; ys = {}
    for rnd in range(5):
        for v in (0, 1):
            ops.lib.call("sgc_set_tuning", b"halo_small", v)
            ys[v] = f()[0].clone()
            t = timed(f)
            if rnd: ts[v].append(t)
    scale = float(ys[0].abs().max())
    print(f"{name:22s} tile kernel {sorted(ts[0])[len(ts[0]) // 2]:6.1f} us | whole-grid brick {sorted(ts[1])[len(ts[1]) // 2]:6.1f} us | max diff {float((ys[0] - ys[1]).abs().max()) / scale:.1e} of the scale", flush=True)
ops.lib.call("sgc_set_tuning", b"halo_small", 1)
