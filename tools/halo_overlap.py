"""Do two halo convolutions from different streams pack the chip?  The 90-GF layer (200 workgroups, one per CU, 256 CUs) back to
back on one stream against the same launches spread over 2 / 4 streams: ideal packing of 200-workgroup kernels is 256 / 200 = 1.28x."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd import ext
ops = ext.ops()
Cin = Cout = 256; g = (40, 40, 16)
xs = [torch.randn(g[0] * g[1] * g[2], Cin, device="cuda") for _ in range(4)]
wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
wh, wl = ops.split_bf16(wt)
streams = [torch.cuda.Stream() for _ in range(4)]
def run(n_streams, n=48):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams[:n_streams]: s.wait_event(e0)
    for i in range(n):
        with torch.cuda.stream(streams[i % n_streams]):
            ops.conv3d_cl_bf16x3(xs[i % 4], wh, wl, g, 3, 1, False, sc, sh, None, True)
    for s in streams[:n_streams]: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(4):
    print(f"round {rnd}: " + " | ".join(f"{k} stream(s) {run(k):6.1f} us/launch" for k in (1, 2, 4)), flush=True)
