"""Where the halo convolution's tap loop spends its time (the 90-GF layer 256 -> 256 at 40x40x16; SGC_HALO_STAGGER=0: the lockstep form): the product
library against timing builds with parts of the loop removed (SGC_HALO_SKIP, csrc/diag.hpp; their results are garbage):
  for m in 1 2 4 8 16 6 24 31; do bash tools/diag_build.sh skip$m conv3d.hip -DSGC_HALO_SKIP=$m; done
Alternated rounds in one process; the first round is the cold one."""
import glob, os, re, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext
libs = {"product": ext.ops()}
for f in sorted(glob.glob(os.path.join(ROOT, "tools/diag/libsgc_skip*.so")), key=lambda f: int(re.findall(r"skip(\d+)", f)[0])):
    libs[re.findall(r"(skip\d+)", f)[0]] = TensorOps(Library(f), "cuda")
for o in libs.values():
    o.lib.call("sgc_set_tuning", b"halo_stagger", int(os.environ.get("SGC_HALO_STAGGER", "1")))
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
for rnd in range(5):
    print(f"round {rnd}: " + " | ".join(f"{nm} {timed(ops):6.1f}" for nm, ops in libs.items()), flush=True)
