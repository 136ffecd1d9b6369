"""Weight layout passes of the training step (sgc_pack_conv_weight both forms, sgc_unpack_conv_wgrad) on the parameter shapes of the
config-2 neck / head / level Linears: us per launch and GB/s of the bytes each pass has to move (read fp32 + write hi | lo = 8 bytes
per element for a pack, 8 for an unpack)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext

ops = ext.ops()
SHAPES = [(1024, 1024, 27), (1024, 512, 27), (512, 512, 27), (512, 256, 27), (256, 256, 27), (128, 256, 27), (128, 512, 27), (128, 1024, 27),
          (28, 128, 27), (1024, 512, 8), (512, 256, 8), (1024, 512, 1), (512, 256, 1), (1280, 256, 1), (256, 256, 1)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"pack": 0.0, "pack_T": 0.0, "unpack": 0.0}
print(f"{'A x B x T':>18} {'pack us':>9} {'GB/s':>7} {'pack^T us':>10} {'GB/s':>7} {'unpack us':>10} {'GB/s':>7}")
for A, B, T in SHAPES:
    w = torch.randn(A, B, T, device="cuda")
    gb = A * B * T * 8 / 1e9
    t0 = timed(lambda: ops.pack_conv_weight(w, pad_rows=4))
    t1 = timed(lambda: ops.pack_conv_weight(w, transpose=True, flip=T == 27, pad_cols=32))
    dw = torch.randn(T, A, B, device="cuda")
    t2 = timed(lambda: ops.unpack_conv_wgrad(dw, (A, B, T)))
    tot["pack"] += t0; tot["pack_T"] += t1; tot["unpack"] += t2
    print(f"{A:>6} x{B:>5} x{T:>3} {t0:9.1f} {gb / t0 * 1e6:7.0f} {t1:10.1f} {gb / t1 * 1e6:7.0f} {t2:10.1f} {gb / t2 * 1e6:7.0f}")
print("sum over the listed shapes (us):", {k: round(v, 1) for k, v in tot.items()})
