"""Where the tile implicit GEMM's time goes on the neck's non-halo layers: the product library against diagnostic builds without
global loads / without MFMAs / without the epilogue (bash tools/diag_build.sh ig_noloads conv3d.hip -DSGC_DIAG_IG_NO_LOADS, ... ;
results of those builds are garbage)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext
libs = {"product": ext.ops()}
for n in ("ig_noloads", "ig_nomfma", "ig_noepi", "nosplit"):
    f = os.path.join(ROOT, f"tools/diag/libsgc_{n}.so")
    if os.path.exists(f):
        libs[n] = TensorOps(Library(f), "cuda")
layers = [("down1.conv1 256->512 s2", 256, 512, (40, 40, 16), 3, 2, False), ("down2.conv1 512->1024 s2", 512, 1024, (20, 20, 8), 3, 2, False),
          ("down2.conv2 1024->1024 @10x10x4", 1024, 1024, (10, 10, 4), 3, 1, False), ("out2 1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1, False),
          ("up2.convT 1024->512", 1024, 512, (10, 10, 4), 2, 2, True), ("up1.convT 512->256", 512, 256, (20, 20, 8), 2, 2, True)]
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Cin, Cout, g, k, s, tr in layers:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); taps = 8 if tr else k ** 3
    wt = torch.randn(taps, Cout, Cin, device="cuda") * 0.01
    sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    wh, wl = libs["product"].split_bf16(wt)
    line = []
    for rnd in range(2):
        for nm, ops in libs.items():
            t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, tr, sc, sh, None, True))
            line.append(f"{nm} {t:6.1f}")
    print(f"{name:34s} " + " | ".join(line), flush=True)
