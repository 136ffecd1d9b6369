"""Tile implicit GEMM on the neck's non-halo layers under the values of one tuning knob, alternated, results compared bit for bit:
python tools/igemm_tall_ab.py [values] [knob] [default]   (igemm_tall 0 / 1 = 128 x 128 against 256 x 128 tiles; conv_waves 8 / 4)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd import ext
ops = ext.ops()
layers = [("down1.conv1 256->512 s2", 256, 512, (40, 40, 16), 3, 2, False), ("down2.conv1 512->1024 s2", 512, 1024, (20, 20, 8), 3, 2, False),
          ("down2.conv2 1024->1024 @10x10x4", 1024, 1024, (10, 10, 4), 3, 1, False), ("out2 1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1, False),
          ("up2.convT 1024->512", 1024, 512, (10, 10, 4), 2, 2, True), ("up1.convT 512->256", 512, 256, (20, 20, 8), 2, 2, True)]
modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
KNOB = (sys.argv[2] if len(sys.argv) > 2 else "igemm_tall").encode()
DEFAULT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Cin, Cout, g, k, s, tr in layers:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); taps = 8 if tr else k ** 3
    wt = torch.randn(taps, Cout, Cin, device="cuda") * 0.01
    sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    line, ref = [], None
    for rnd in range(3):
        for m in modes:
            ops.lib.call("sgc_set_tuning", KNOB, m)
            t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, tr, sc, sh, None, True))
            y = ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, tr, sc, sh, None, True)[0]
            ref = y if ref is None else ref
            assert torch.equal(y, ref), (name, m)
            line.append(f"m{m} {t:6.1f}")
    print(f"{name:34s} " + " | ".join(line), flush=True)
ops.lib.call("sgc_set_tuning", KNOB, DEFAULT)
