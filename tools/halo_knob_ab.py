"""Halo convolution: values of one tuning knob alternated in one process (4 rounds x 30 launches per value, HIP events), results
compared bit for bit, on the 3x3x3 stride-1 layers of the config-2 / config-4 necks.
Usage: python tools/halo_knob_ab.py <knob> <v0,v1,...>     e.g.  halo_stagger 0,1   (0 = lockstep, 1 = staggered waves)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd import ext
ops = ext.ops()
layers = [("256->256 @40x40x16", 256, 256, (40, 40, 16)), ("256->128 @40x40x16", 256, 128, (40, 40, 16)), ("128->28 @40x40x16", 128, 28, (40, 40, 16)),
          ("512->512 @20x20x8", 512, 512, (20, 20, 8)), ("512->128 @20x20x8", 512, 128, (20, 20, 8)), ("128->128 @80x80x32", 128, 128, (80, 80, 32))]
knob = (sys.argv[1] if len(sys.argv) > 1 else "halo_stagger").encode()
values = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
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
    ts, ref = {m: [] for m in values}, None
    for rnd in range(5):
        for m in values:
            ops.lib.call("sgc_set_tuning", knob, m)
            t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True))
            y = ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)[0]
            ref = y if ref is None else ref
            assert torch.equal(y, ref), (name, m, float((y - ref).abs().max()))
            if rnd:                                   # the first round is the cold one
                ts[m].append(t)
    fl = 2 * Cin * Cout * V * 27
    print(f"{name:22s} " + " | ".join(f"{knob.decode()}={m} {sorted(ts[m])[len(ts[m]) // 2]:6.1f} us ({fl * 3 / sorted(ts[m])[len(ts[m]) // 2] / 1e6 / 2.5e3:.3f} of 2.5 PF issued)" for m in values)
          + "  bit-identical", flush=True)
ops.lib.call("sgc_set_tuning", knob, values[-1])
