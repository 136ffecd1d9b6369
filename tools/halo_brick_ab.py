"""Brick shape of the halo convolution (4x4x16 / 4x8x8 / 8x8x4) on the 40x40x16 layers: interleaved A/B, settled clocks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Cin, Cout, g in [("256->256 @40x40x16", 256, 256, (40, 40, 16)), ("256->128 @40x40x16", 256, 128, (40, 40, 16)), ("512->512 @20x20x8", 512, 512, (20, 20, 8))]:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
    sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    f = lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    ref = None
    for rnd in range(5):
        line = []
        for b, nb in ((3, 1), (1, 1), (2, 1), (2, 0)):
            ops.lib.call("sgc_set_tuning", b"halo_brick", b)
            ops.lib.call("sgc_set_tuning", b"halo_stagger", nb)
            y, _ = f()
            if ref is None: ref = y.clone()
            line.append(f"brick{b}/stagger{nb} {timed(f):6.1f} us same={bool(torch.equal(y, ref))}")
        print(name, "round", rnd, " | ".join(line), flush=True)
ops.lib.call("sgc_set_tuning", b"halo_brick", 0)
ops.lib.call("sgc_set_tuning", b"halo_stagger", 1)
