"""The V projection of the projected-query attention, head by head (sgc_linear_rows_blockdiag_bf16x3) against the dense 8C -> C GEMM
with 7/8 zero blocks it replaces: alternated rounds, HIP events, median; bytes = the x read + the y write."""
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


for rows, K in [(73651, 128), (47000, 128), (9216, 128), (4608, 128), (6400, 256), (1600, 256), (400, 256)]:
    G, Nh = 8, K // 8
    x = torch.randn(rows, G * K, device="cuda")
    w = torch.randn(G, Nh, K, device="cuda") * (1.0 / K ** 0.5)
    b = torch.randn(G * Nh, device="cuda") * 0.1
    hi, lo = ops.split_bf16(w)
    dense = torch.zeros(G * Nh, G * K, device="cuda")
    for h in range(G):
        dense[h * Nh:(h + 1) * Nh, h * K:(h + 1) * K] = w[h]
    dhi, dlo = ops.split_bf16(dense.view(1, G * Nh, G * K))
    f_b = lambda: ops.linear_rows_blockdiag(x, hi, lo, b)
    f_d = lambda: ops.linear_rows_bf16x3(x, dhi, dlo, b)
    same = bool(torch.equal(f_b(), f_d()))
    tb, td = [], []
    for rnd in range(5):
        t1, t2 = timed(f_b), timed(f_d)
        if rnd: tb.append(t1); td.append(t2)
    tb, td = sorted(tb)[len(tb) // 2], sorted(td)[len(td) // 2]
    mb = rows * (G * K + G * Nh) * 4 / 1e6
    print(f"{rows:6d} x {G * K} -> {G * Nh}: head by head {tb:6.1f} us ({mb / tb:4.2f} TB/s of {mb:.0f} MB) | dense {td:6.1f} us | bit-identical {same}", flush=True)
