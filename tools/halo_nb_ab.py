"""Halo convolution: staged weights (`halo_nb` 2) against the direct-B form (0), alternated in one process, results compared bit
for bit, on the config-2 neck's 3x3x3 stride-1 layers."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd import ext
ops = ext.ops()
layers = [("256->256 @40x40x16", 256, 256, (40, 40, 16)), ("256->128 @40x40x16", 256, 128, (40, 40, 16)), ("512->512 @20x20x8", 512, 512, (20, 20, 8)),
          ("512->128 @20x20x8", 512, 128, (20, 20, 8)), ("128->128 @80x80x32", 128, 128, (80, 80, 32))]
modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "2,0").split(",")]
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
    wh, wl = ops.split_bf16(wt)
    line, ref = [], None
    for rnd in range(4):
        for m in modes:
            ops.lib.call("sgc_set_tuning", b"halo_nb", m)
            t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True))
            y = ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)[0]
            ref = y if ref is None else ref
            assert torch.equal(y, ref), (name, m, float((y - ref).abs().max()))
            line.append(f"nb{m} {t:6.1f}")
    print(f"{name:22s} " + " | ".join(line), flush=True)
ops.lib.call("sgc_set_tuning", b"halo_nb", 2)
