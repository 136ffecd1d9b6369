"""Where the halo weight-gradient kernel (SGC_WGRAD_HALO=1 double-buffered, 2 single-buffered) spends its time (256 -> 256 at 40x40x16): the product library against timing
builds with parts removed (SGC_WGRAD_SKIP, csrc/diag.hpp; their results are garbage):
  for m in 1 2 4 6 7 8 16 24 30; do bash tools/diag_build.sh wskip$m conv3d.hip -DSGC_WGRAD_SKIP=$m; done
Alternated rounds in one process; the first round is the cold one."""
import glob, os, re, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext
libs = {"product": ext.ops()}
for f in sorted(glob.glob(os.path.join(ROOT, "tools/diag/libsgc_wskip*.so")), key=lambda f: int(re.findall(r"wskip(\d+)", f)[0])):
    libs[re.findall(r"(wskip\d+)", f)[0]] = TensorOps(Library(f), "cuda")
for o in libs.values():
    o.lib.call("sgc_set_tuning", b"wgrad_halo", int(os.environ.get("SGC_WGRAD_HALO", "1")))
Cin = Cout = int(os.environ.get("C", "256")); g = (40, 40, 16)
x = torch.randn(g[0] * g[1] * g[2], Cin, device="cuda")
dy = torch.randn(g[0] * g[1] * g[2], Cout, device="cuda")
def timed(ops, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv3d_wgrad_bf16x3(x, dy, g, 3, 1); e0.record()
    for _ in range(n): ops.conv3d_wgrad_bf16x3(x, dy, g, 3, 1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(4):
    print(f"round {rnd}: " + " | ".join(f"{nm} {timed(ops):6.1f}" for nm, ops in libs.items()), flush=True)
