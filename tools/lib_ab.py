"""Two builds of the library against each other on the halo layers, alternated in one process (same box, same clocks):
python tools/lib_ab.py tools/diag/libsgc_prev.so   -- the argument is the OTHER library (e.g. the previous commit's build)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext
libs = {"this": ext.ops(), "other": TensorOps(Library(os.path.join(ROOT, sys.argv[1])), "cuda")}
layers = [("256->256 @40x40x16", 256, 256, (40, 40, 16)), ("256->128 @40x40x16", 256, 128, (40, 40, 16)), ("512->512 @20x20x8", 512, 512, (20, 20, 8)),
          ("128->128 @80x80x32", 128, 128, (80, 80, 32))]
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Cin, Cout, g in layers:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda")
    wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.randn(Cout, device="cuda")
    wh, wl = libs["this"].split_bf16(wt)
    line, ref = [], None
    for rnd in range(5):
        for nm, ops in libs.items():
            t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True))
            y = ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)[0]
            ref = y if ref is None else ref
            assert torch.equal(y, ref), (name, nm)
            line.append(f"{nm} {t:6.1f}")
    print(f"{name:22s} " + " | ".join(line), flush=True)
