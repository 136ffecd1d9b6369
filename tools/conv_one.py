"""One conv layer in a loop (for rocprofv3 --pmc passes): python3 tools/conv_one.py Cin Cout gx gy gz [k s reps [wz]]
(a 9th argument 1 = the Winograd-z form of a 3x3x3 stride-1 layer, sgc_conv3d_winograd_z_bf16x3)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
a = [int(v) for v in sys.argv[1:]]
Cin, Cout, g = a[0], a[1], tuple(a[2:5])
k = a[5] if len(a) > 5 else 3
s = a[6] if len(a) > 6 else 1
reps = a[7] if len(a) > 7 else 10
wz = len(a) > 8 and a[8] == 1
x = torch.randn(g[0] * g[1] * g[2], Cin, device="cuda")
wt = torch.randn(k ** 3, Cout, Cin, device="cuda") * 0.01
wh, wl = ops.split_bf16(wt)
if wz:
    gh, gl = ops.split_operand(ops.winograd_z_weights(wt))
for _ in range(reps):
    if wz:
        ops.conv3d_winograd_z(x, gh, gl, g)
    else:
        ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, False)
torch.cuda.synchronize()
