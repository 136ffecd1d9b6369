"""Prices a Winograd F(2,3)-along-z form of the 90-GF layer (256 -> 256 @ 40x40x16) from MEASURED parts, without building it
(DESIGN.md 7.1).  In the two-pass form every transform-domain position k is a 3x3 convolution over (x, y) on a stack of Z/2
"images" with its own weights G_k -- the geometry of sgc_conv2d_nhwc_bf16x3 on bricks of 4 images x 8 x 8 pixels (`halo_2d` = 2):
  (a) ONE position: 8 images of 40x40, 256 -> 256 (100 workgroups);
  (b) ALL FOUR as one launch: 32 images (400 workgroups; shared weights here -- the real form reads four weight sets, the same
      bytes per workgroup);
  (c) the elementwise passes around it, priced by copies of the same byte counts: input transform (read V*C, write 2*V*C),
      output transform + epilogue (read 2*V*Cout (+ V*Cout residual), write V*Cout);
against (d) the direct 3x3x3 layer as shipped.  Results of (a)/(b) are NOT a convolution of anything meaningful: timing only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext

ops = ext.ops()


def timed(fn, n=30, rounds=4):
    ts = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


C = Co = 256
g3 = (40, 40, 16)
V = g3[0] * g3[1] * g3[2]
x3 = torch.randn(V, C, device="cuda")
w27 = torch.randn(27, Co, C, device="cuda") * 0.01
h27, l27 = ops.split_bf16(w27)
w9 = torch.randn(9, Co, C, device="cuda") * 0.01
h9, l9 = ops.split_bf16(w9)
sc, sh = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
res = torch.randn(V, Co, device="cuda")
t_direct = timed(lambda: ops.conv3d_cl_bf16x3(x3, h27, l27, g3, 3, 1, False, sc, sh, res, 1))
print(f"(d) direct 3x3x3 layer as shipped                      {t_direct:7.1f} us")
ops.lib.call("sgc_set_tuning", b"halo_2d", 2)
for name, n_img in (("(a) one position: 8 images of 40x40, 100 workgroups  ", 8), ("(b) four positions as one launch: 32 images, 400 wg  ", 32)):
    xi = torch.randn(n_img * 1600, C, device="cuda")
    t = timed(lambda: ops.conv2d_nhwc_bf16x3(xi, h9, l9, (n_img, 40, 40), 3))
    print(f"{name} {t:7.1f} us")
    if n_img == 32:
        t_b = t
ops.lib.call("sgc_set_tuning", b"halo_2d", 1)
# elementwise passes: the same bytes through the chip (torch elementwise kernels run at the copy rate)
a = torch.randn(V, C, device="cuda"); t2 = torch.empty(2 * V, C, device="cuda")
t_in = timed(lambda: torch.add(a.repeat(2, 1), 1.0, out=t2)) if False else timed(lambda: (t2[:V].copy_(a), t2[V:].copy_(a)))
m = torch.randn(2 * V, Co, device="cuda"); y = torch.empty(V, Co, device="cuda")
t_out = timed(lambda: torch.add(torch.add(m[:V], m[V:]), res, out=y))
print(f"(c) input transform priced as read V*C + write 2*V*C   {t_in:7.1f} us;  output transform + epilogue priced as read 2*V*Co + V*Co, write V*Co {t_out:7.1f} us")
print(f"two-pass Winograd F(2,3)-z, sum of measured parts      {t_b + t_in + t_out:7.1f} us   against {t_direct:.1f} us direct   (stop rule of the round-4 review: <= 175 us)")
