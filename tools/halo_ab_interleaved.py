import os, sys, time, torch
sys.path.insert(0, ".")
from sgcdet_amd import ext
ops = ext.ops()
g = (40, 40, 16)
x = torch.randn(25600, 256, device="cuda"); wt = torch.randn(27, 256, 256, device="cuda") * 0.01
sc = torch.ones(256, device="cuda"); sh = torch.zeros(256, device="cuda")
wh, wl = ops.split_bf16(wt)
res = {0: [], 2: []}
for r in range(8):
    for v in (0, 2):
        ops.lib.call("sgc_set_tuning", b"halo_ring", v)
        for _ in range(3): ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t) / 40 * 1e6)
for v in (0, 2):
    print("halo_ring", v, " ".join(f"{t:6.1f}" for t in res[v]), " median %.1f" % sorted(res[v])[4])
