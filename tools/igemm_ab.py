"""Implicit-GEMM kernel A/B on one tuning knob (default igemm_direct: epilogue through LDS vs transposed accumulators with
16-byte stores from registers): Linears over row lists and the convolutions the halo kernel does not take; time and
bit-identity.  SGC_AB_KNOB=<key> SGC_AB_VALUES=0,1"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
KNOB = os.environ.get("SGC_AB_KNOB", "igemm_direct").encode()
VAR = [int(v) for v in os.environ.get("SGC_AB_VALUES", "0,1").split(",")]


def timed(f, n=20):
    for _ in range(3):
        y = f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        y = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6, y


def report(name, res):
    print(f"{name:26s}: " + " | ".join(f"v{g} {res[g][0]:7.1f} us" for g in VAR) +
          f" | identical {all(torch.equal(res[VAR[0]][1], res[g][1]) for g in VAR)}", flush=True)


print("Linears (rows x Cin -> Cout)")
for rows, cin, cout in [(204800, 256, 256), (188800, 256, 256), (77000, 256, 512), (77000, 256, 128), (51200, 256, 256), (6400, 256, 256), (6400, 512, 256)]:
    x = torch.randn(rows, cin, device="cuda"); wt = torch.randn(1, cout, cin, device="cuda") * 0.05
    sh = torch.randn(cout, device="cuda"); wh, wl = ops.split_bf16(wt)
    res = {}
    for g in VAR:
        ops.lib.call("sgc_set_tuning", KNOB, g)
        res[g] = timed(lambda: ops.linear_rows_bf16x3(x, wh, wl, sh))
    report(f"{rows} x {cin} -> {cout}", res)
x = torch.randn(40 * 5120, 256, device="cuda"); wt = torch.randn(1, 256, 256, device="cuda") * 0.05
sh = torch.randn(256, device="cuda"); wh, wl = ops.split_bf16(wt)
res = {}
for g in VAR:
    ops.lib.call("sgc_set_tuning", KNOB, g)
    res[g] = timed(lambda: ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, 40, 5120, 8))
report("head-major 204800 x 256", res)
print("convolutions")
layers = [("256->512 s2 @40x40x16", 256, 512, (40, 40, 16), 3, 2, False), ("512->1024 s2 @20x20x8", 512, 1024, (20, 20, 8), 3, 2, False),
          ("1024->1024 @10x10x4", 1024, 1024, (10, 10, 4), 3, 1, False), ("1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1, False),
          ("convT 1024->512", 1024, 512, (10, 10, 4), 2, 2, True), ("convT 512->256", 512, 256, (20, 20, 8), 2, 2, True),
          ("1x1 s2 256->512", 256, 512, (40, 40, 16), 1, 2, False)]
for name, Cin, Cout, g, k, s, tr in layers:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); taps = 8 if tr else k ** 3
    wt = torch.randn(taps, Cout, Cin, device="cuda") * 0.01
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.randn(Cout, device="cuda")
    r = torch.randn((V * 8 if tr else (V // 8 if s == 2 else V)), Cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    res = {}
    for gl in VAR:
        ops.lib.call("sgc_set_tuning", KNOB, gl)
        res[gl] = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, tr, sc, sh, r, True)[0], 10)
    report(name, res)
