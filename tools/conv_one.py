"""One conv layer in a loop (for rocprofv3 --pmc passes): python3 tools/conv_one.py Cin Cout gx gy gz [k s reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
a = [int(v) for v in sys.argv[1:]]
Cin, Cout, g = a[0], a[1], tuple(a[2:5])
k = a[5] if len(a) > 5 else 3
s = a[6] if len(a) > 6 else 1
reps = a[7] if len(a) > 7 else 10
x = torch.randn(g[0] * g[1] * g[2], Cin, device="cuda")
wt = torch.randn(k ** 3, Cout, Cin, device="cuda") * 0.01
wh, wl = ops.split_bf16(wt)
for _ in range(reps):
    ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, False)
torch.cuda.synchronize()
