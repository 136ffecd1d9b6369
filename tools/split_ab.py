"""interleaved A/B of the implicit-GEMM split target (workgroups a layer is split into) on the layers that split"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
layers = [("256->512 s2 @40x40x16", 256, 512, (40, 40, 16), 3, 2), ("512->1024 s2 @20x20x8", 512, 1024, (20, 20, 8), 3, 2),
          ("1024->1024 @10x10x4", 1024, 1024, (10, 10, 4), 3, 1), ("1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1),
          ("128->28 @10x10x4", 128, 28, (10, 10, 4), 3, 1)]
targets = [int(v) for v in os.environ.get("SGC_SPLIT_TARGETS", "512,256,128").split(",")]
for name, cin, cout, g, k, s in layers:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, cin, device="cuda"); wt = torch.randn(27, cout, cin, device="cuda") * 0.01
    sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    res = {t: [] for t in targets}
    for r in range(6):
        for t in targets:
            ops.lib.call("sgc_set_tuning", b"split_target", t)
            for _ in range(3): ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, False, sc, sh, None, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, False, sc, sh, None, True)
            torch.cuda.synchronize(); res[t].append((time.perf_counter() - t0) / 40 * 1e6)
    print(f"{name:26s} " + " | ".join(f"target {t}: {sorted(res[t])[3]:7.1f} us" for t in targets), flush=True)
