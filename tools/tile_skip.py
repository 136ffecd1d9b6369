"""Where the tile implicit GEMM (conv3d_igemm_bf16x3_kernel) spends its time on the layers the halo kernel cannot take -- the product
library against timing builds with parts of the K loop removed (SGC_TILE_SKIP, csrc/diag.hpp; THEIR RESULTS ARE GARBAGE):
  for m in 1 2 4 8 14 16 32 64; do bash tools/diag_build.sh tskip$m conv3d.hip -DSGC_TILE_SKIP=$m; done
bits: 1 no MFMAs, 2 no input loads, 4 no weight loads, 8 no split + LDS stores, 16 no fragment reads, 32 no barrier per step,
64 no epilogue.  Alternated rounds in one process (the first one is cold), median of the rest."""
import glob, os, re, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext

libs = {"product": ext.ops()}
for f in sorted(glob.glob(os.path.join(ROOT, "tools/diag/libsgc_tskip*.so"))):
    libs[re.findall(r"libsgc_(\w+)\.so", f)[0]] = TensorOps(Library(f), "cuda")
order = ["product"] + sorted((n for n in libs if n != "product"), key=lambda n: int(n[5:]))
LAYERS = [("256->512 s2 @40x40x16", 256, 512, (40, 40, 16), 3, 2, False), ("512->1024 s2 @20x20x8", 512, 1024, (20, 20, 8), 3, 2, False),
          ("1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1, False), ("1024->512 T @10x10x4", 1024, 512, (10, 10, 4), 2, 2, True),
          ("512->256 T @20x20x8", 512, 256, (20, 20, 8), 2, 2, True)]


def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("us per call (kernel + split epilogue), latency geometry; columns: " + " | ".join(order))
for name, Cin, Cout, g, k, s, tr in LAYERS:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda")
    wt = torch.randn(8 if tr else k ** 3, Cout, Cin, device="cuda") * 0.01
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.randn(Cout, device="cuda") * 0.1
    wh, wl = libs["product"].split_bf16(wt)
    ts = {n: [] for n in order}
    for rnd in range(5):
        for n in order:
            t = timed(lambda: libs[n].conv3d_cl_bf16x3(x, wh, wl, g, k, s, tr, sc, sh, None, True))
            if rnd: ts[n].append(t)
    print(f"{name:24s} " + " | ".join(f"{sorted(ts[n])[len(ts[n]) // 2]:6.1f}" for n in order), flush=True)
