"""Where the halo convolution's time goes (the 90-GF layer 256 -> 256 at 40x40x16, warm): the product library against diagnostic
builds without the weight loads / without the weight ds_writes / without the barrier per tap / without all three
(bash tools/diag_build.sh h_nobload conv3d.hip -DSGC_DIAG_HALO_NO_BLOAD, ...; results of those builds are garbage).
Interleaved rounds; the first round is the cold one."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext
libs = {"product": ext.ops()}
for n in ("h_nobload", "h_nobwrite", "h_nobar", "h_noboth"):
    f = os.path.join(ROOT, f"tools/diag/libsgc_{n}.so")
    if os.path.exists(f):
        libs[n] = TensorOps(Library(f), "cuda")
Cin = Cout = 256; g = (40, 40, 16)
x = torch.randn(g[0] * g[1] * g[2], Cin, device="cuda")
wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
wh, wl = libs["product"].split_bf16(wt)
def timed(ops, n=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True); e0.record()
    for _ in range(n): ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(6):
    print(f"round {rnd}: " + " | ".join(f"{nm} {timed(ops):6.1f}" for nm, ops in libs.items()), flush=True)
