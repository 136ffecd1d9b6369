"""Timing-only ablations of the persistent row GEMM (SGC_DIAG=1): where does a tile's time go?"""
import os, sys, torch
os.environ["SGC_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, cin, cout, hm in [(204800, 256, 256, True), (204800, 256, 256, False), (76856, 256, 512, False), (76856, 256, 128, False)]:
    x = torch.randn(rows, cin, device="cuda"); wt = torch.randn(1, cout, cin, device="cuda") * 0.05
    sh = torch.randn(cout, device="cuda"); wh, wl = ops.split_bf16(wt)
    y = torch.empty(rows, cout, device="cuda")
    fn = (lambda: ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, 40, 5120, 8)) if hm else (lambda: ops.linear_rows_bf16x3(x, wh, wl, sh, out=y))
    for rnd in range(2):
        line = []
        for d, name in [(0, "full"), (1, "no stores"), (2, "no loads"), (4, "no mfma"), (3, "no loads+stores"), (5, "no stores+mfma"), (6, "no loads+mfma"), (7, "none")]:
            ops.lib.call("sgc_set_tuning", b"rows_diag", d)
            line.append(f"{name} {timed(fn):6.1f}")
        print(f"{rows} x {cin} -> {cout} hm={hm}: " + " | ".join(line), flush=True)
ops.lib.call("sgc_set_tuning", b"rows_diag", 0)
