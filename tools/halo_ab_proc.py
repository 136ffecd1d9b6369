"""warm per-process timing of the 90-GF halo layer (8 rounds x 40 launches): SGC_DIAG_LIB selects a diagnostic build"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
g = (40, 40, 16)
x = torch.randn(25600, 256, device="cuda"); wt = torch.randn(27, 256, 256, device="cuda") * 0.01
sc = torch.ones(256, device="cuda"); sh = torch.zeros(256, device="cuda")
wh, wl = ops.split_bf16(wt)
ts = []
for r in range(8):
    for _ in range(3): ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 40 * 1e6)
print(os.environ.get("SGC_DIAG_LIB", "product"), " ".join(f"{t:6.1f}" for t in ts), " median %.1f" % sorted(ts)[4])
