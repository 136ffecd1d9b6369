"""Interleaved A/B of one tuning knob over conv / Linear shapes (6 rounds x 40 launches per value, median of the rounds):
python tools/knob_ab.py <knob> <v1,v2,...> [linear|conv|halo]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
knob, values = sys.argv[1].encode(), [int(v) for v in sys.argv[2].split(",")]
which = sys.argv[3] if len(sys.argv) > 3 else "conv"
conv = [("256->256 @40x40x16", 256, 256, (40, 40, 16), 3, 1), ("256->128 @40x40x16", 256, 128, (40, 40, 16), 3, 1),
        ("128->28 @40x40x16", 128, 28, (40, 40, 16), 3, 1), ("512->512 @20x20x8", 512, 512, (20, 20, 8), 3, 1),
        ("512->128 @20x20x8", 512, 128, (20, 20, 8), 3, 1), ("1024->1024 @10x10x4", 1024, 1024, (10, 10, 4), 3, 1),
        ("1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1), ("256->512 s2 @40x40x16", 256, 512, (40, 40, 16), 3, 2),
        ("512->1024 s2 @20x20x8", 512, 1024, (20, 20, 8), 3, 2), ("1x1 s2 256->512", 256, 512, (40, 40, 16), 1, 2)]
linear = [(204800, 256, 256), (77000, 256, 512), (77000, 256, 128), (51200, 256, 256), (6400, 256, 256), (6400, 512, 256)]


def ab(name, f):
    res = {v: [] for v in values}
    for r in range(6):
        for v in values:
            ops.lib.call("sgc_set_tuning", knob, v)
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): f()
            torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 40 * 1e6)
    print(f"{name:26s} " + " | ".join(f"{v}: {sorted(res[v])[3]:7.1f} us" for v in values), flush=True)


if which in ("conv", "halo"):
    for name, cin, cout, g, k, s in (conv[:7] if which == "halo" else conv):
        V = g[0] * g[1] * g[2]
        x = torch.randn(V, cin, device="cuda"); wt = torch.randn(k ** 3, cout, cin, device="cuda") * 0.01
        sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
        wh, wl = ops.split_bf16(wt)
        ab(name, lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, False, sc, sh, None, True))
else:
    for rows, cin, cout in linear:
        x = torch.randn(rows, cin, device="cuda"); wt = torch.randn(1, cout, cin, device="cuda") * 0.05
        sh = torch.randn(cout, device="cuda"); wh, wl = ops.split_bf16(wt)
        ab(f"{rows} x {cin} -> {cout}", lambda: ops.linear_rows_bf16x3(x, wh, wl, sh))
