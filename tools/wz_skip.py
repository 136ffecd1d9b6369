"""Where the Winograd-z form of the halo convolution (sgc_conv3d_winograd_z_bf16x3: the halo kernel's 2-D mode on the virtual
stack of 4 Z/2 images + the output transform) spends its time -- the product library against timing builds with parts of the
9-tap loop removed (SGC_HALO_SKIP, csrc/diag.hpp; THEIR RESULTS ARE GARBAGE) and against other builds of the same kernel
(tools/diag/libsgc_<name>.so that are not skip builds are checked bit for bit):
  for m in 1 2 6 24 32 64 128 192; do bash tools/diag_build.sh skip$m conv3d.hip -DSGC_HALO_SKIP=$m; done
Alternated rounds in one process; the first round is the cold one.  Usage: python tools/wz_skip.py [Cin Cout gx gy gz]"""
import glob, os, re, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext

a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [256, 256, 40, 40, 16]
Cin, Cout, g = a[0], a[1], tuple(a[2:5])
libs = {"product": ext.ops()}
for f in sorted(glob.glob(os.path.join(ROOT, "tools/diag/libsgc_*.so"))):
    nm = re.findall(r"libsgc_(\w+)\.so", f)[0]
    libs[nm] = TensorOps(Library(f), "cuda")
order = ["product"] + sorted((n for n in libs if n != "product" and not n.startswith("skip"))) + \
    sorted((n for n in libs if n.startswith("skip")), key=lambda n: int(n[4:]))
V = g[0] * g[1] * g[2]
x = torch.randn(V, Cin, device="cuda")
wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.randn(Cout, device="cuda")
gh, gl = libs["product"].split_operand(libs["product"].winograd_z_weights(wt))
wh, wl = libs["product"].split_bf16(wt)
y = torch.empty(V, Cout, device="cuda")


def timed(ops, n=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv3d_winograd_z(x, gh, gl, g, sc, sh, None, True, out=y); e0.record()
    for _ in range(n):
        ops.conv3d_winograd_z(x, gh, gl, g, sc, sh, None, True, out=y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = libs["product"].conv3d_winograd_z(x, gh, gl, g, sc, sh, None, True)[0].clone()
for nm in order:
    if not nm.startswith("skip"):
        got = libs[nm].conv3d_winograd_z(x, gh, gl, g, sc, sh, None, True)[0]
        print(f"{nm}: bit-identical to the product = {bool(torch.equal(got, ref))}", flush=True)
print(f"layer {Cin} -> {Cout} @ {g}, Winograd-z form (both launches), us per call; skip bits: 1 no barrier per tap, 2 no weight ds_write, "
      "4 no weight load, 8 weight fragments read once, 16 halo fragments read once, 32 no MFMAs, 64 no halo restaging per slice, 128 no epilogue")
for rnd in range(5):
    print(f"round {rnd}: " + " | ".join(f"{nm} {timed(libs[nm]):6.1f}" for nm in order), flush=True)
# the direct form on the same layer, for the ratio
def timed_direct(n=30):
    ops = libs["product"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True); e0.record()
    for _ in range(n):
        ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("direct form, product: " + " ".join(f"{timed_direct():6.1f}" for _ in range(3)))
