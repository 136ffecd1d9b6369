"""Fixed cost of a halo-convolution launch: time of Cin -> 256 at 40x40x16 for Cin = 32 .. 256 (1 .. 8 channel slices per workgroup,
200 workgroups each time): the slope is a slice, the intercept is launch + prologue + epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
ops.lib.call("sgc_set_tuning", b"halo_split_target", 1)      # never split the channel slices
def timed(fn, n=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = (40, 40, 16); V = 25600; Cout = 256
res = {}
for rnd in range(4):
    for Cin in (32, 64, 128, 192, 256):
        x = torch.randn(V, Cin, device="cuda"); wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
        sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
        wh, wl = ops.split_bf16(wt)
        t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True))
        if rnd: res.setdefault(Cin, []).append(t)
med = {c: sorted(v)[len(v) // 2] for c, v in res.items()}
print(" ".join(f"Cin={c}: {t:.1f} us" for c, t in med.items()))
slope = (med[256] - med[64]) / 6.0
print(f"per slice {slope:.2f} us (MFMA-issue floor 18.0 at 2.3 GHz), intercept {med[256] - 8 * slope:.1f} us")
