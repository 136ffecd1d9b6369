"""A/B of the persistent weight-stationary row GEMM (csrc/rows_gemm.hip, knob rows_gemm=1) against the tile-per-workgroup
implicit-GEMM kernel (rows_gemm=0): bit equality of the two (same k order, same product order), error against float64,
device-side row counts, the head-major / residual / relu epilogues, and interleaved timing (6 rounds x 20 launches).

    python tools/rows_gemm_check.py [--quick]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext  # noqa: E402

ops = ext.ops()


def knob(v):
    ops.lib.call("sgc_set_tuning", b"rows_gemm", int(v))


def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, fn, rounds=6):
    res = {0: [], 1: []}
    for _ in range(rounds):
        for g in (0, 1):
            knob(g)
            res[g].append(timed(fn))
    knob(1)
    med = {g: sorted(v)[len(v) // 2] for g, v in res.items()}
    print(f"{name:34s} old {med[0]:8.1f} us (min {min(res[0]):7.1f})   new {med[1]:8.1f} us (min {min(res[1]):7.1f})   x{med[0] / med[1]:.2f}",
          flush=True)
    return med


def main():
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    bad = 0
    shapes = [(204800, 256, 256), (76856, 256, 128), (76856, 256, 512), (51200, 256, 256), (12800, 256, 256), (6400, 256, 256),
              (6400, 256, 512), (800, 256, 256), (400, 256, 256), (33, 256, 128), (1, 256, 256), (31, 128, 128),
              (40000, 128, 128), (40000, 128, 256), (9009, 256, 512)]
    for rows, cin, cout in shapes:
        x = torch.randn(rows, cin, device="cuda")
        wt = torch.randn(1, cout, cin, device="cuda") * 0.05
        sh = torch.randn(cout, device="cuda")
        wh, wl = ops.split_bf16(wt)
        knob(0)
        y0 = ops.linear_rows_bf16x3(x, wh, wl, sh)
        knob(1)
        y1 = ops.linear_rows_bf16x3(x, wh, wl, sh)
        ref = (x[:4096].double() @ wt[0].double().t() + sh.double())
        err = float((y1[:4096].double() - ref).abs().max() / ref.abs().max())
        same = bool(torch.equal(y0, y1))
        # device-side count: rows past it untouched
        cnt = max(1, rows * 2 // 3)
        cdev = torch.tensor([cnt], dtype=torch.int32, device="cuda")
        out = torch.full((rows, cout), 7.0, device="cuda")
        ops.linear_rows_bf16x3(x, wh, wl, sh, count=cdev, out=out)
        ok_cnt = bool(torch.equal(out[:cnt], y1[:cnt])) and bool((out[cnt:] == 7.0).all())
        print(f"{rows:7d} x {cin} -> {cout}: new == old {same}   rel err vs f64 {err:.2e}   count contract {ok_cnt}", flush=True)
        bad += (not same) + (not ok_cnt) + (err > 1e-4)
    # head-major, fp32 and bf16
    for N, S, M, cin in [(40, 5120, 8, 256), (3, 47, 8, 256), (5, 333, 8, 128), (2, 32, 8, 256)]:
        cout = cin
        x = torch.randn(N * S, cin, device="cuda")
        wt = torch.randn(1, cout, cin, device="cuda") * 0.05
        sh = torch.randn(cout, device="cuda")
        wh, wl = ops.split_bf16(wt)
        for dt in (torch.float32, torch.bfloat16):
            knob(0)
            y0 = ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, N, S, M, out_dtype=dt)
            knob(1)
            y1 = ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, N, S, M, out_dtype=dt)
            rows = ops.linear_rows_bf16x3(x, wh, wl, sh)
            perm = rows.view(N, S, M, cout // M).permute(0, 2, 1, 3).contiguous()
            same = bool(torch.equal(y0, y1))
            okp = bool(torch.equal(y1.float(), perm.to(dt).float()))
            print(f"head-major N={N} S={S} C={cin} {dt}: new == old {same}   == permuted rows {okp}", flush=True)
            bad += (not same) + (not okp)
    # the FFN's two calls: relu epilogue, residual epilogue
    for rows in (6400, 77):
        x = torch.randn(rows, 256, device="cuda")
        w1 = torch.randn(1, 512, 256, device="cuda") * 0.05
        b1 = torch.randn(512, device="cuda")
        sc = torch.rand(512, device="cuda") + 0.5
        res = torch.randn(rows, 512, device="cuda")
        h1, l1 = ops.split_bf16(w1)
        for relu, r in ((2, None), (0, res), (1, res), (2, res), (0, None)):
            knob(0)
            y0, _ = ops.conv3d_cl_bf16x3(x, h1, l1, (rows, 1, 1), 1, 1, False, sc, b1, r, relu)
            knob(1)
            y1, _ = ops.conv3d_cl_bf16x3(x, h1, l1, (rows, 1, 1), 1, 1, False, sc, b1, r, relu)
            same = bool(torch.equal(y0, y1))
            print(f"1x1x1 conv rows={rows} relu={relu} residual={r is not None}: new == old {same}", flush=True)
            bad += not same
    print("MISMATCHES:", bad, flush=True)
    if quick:
        return bad
    for rows, cin, cout in [(204800, 256, 256), (76856, 256, 128), (76856, 256, 512), (51200, 256, 256), (12800, 256, 256),
                            (6400, 256, 256), (6400, 256, 512), (800, 256, 256), (400, 256, 256), (9009, 256, 512), (9009, 256, 128),
                            (250000, 128, 128), (800000, 128, 128)]:
        x = torch.randn(rows, cin, device="cuda")
        wt = torch.randn(1, cout, cin, device="cuda") * 0.05
        sh = torch.randn(cout, device="cuda")
        wh, wl = ops.split_bf16(wt)
        y = torch.empty(rows, cout, device="cuda")
        ab(f"{rows} x {cin} -> {cout}", lambda: ops.linear_rows_bf16x3(x, wh, wl, sh, out=y))
    x = torch.randn(204800, 256, device="cuda")
    wt = torch.randn(1, 256, 256, device="cuda") * 0.05
    sh = torch.randn(256, device="cuda")
    wh, wl = ops.split_bf16(wt)
    ab("head-major 40 x 5120 x 256 f32", lambda: ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, 40, 5120, 8))
    ab("head-major 40 x 5120 x 256 bf16", lambda: ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, 40, 5120, 8, out_dtype=torch.bfloat16))
    y = torch.empty(204800, 256, device="cuda")
    for rnd in range(3):
        for d in (0, 1, 2):
            ops.lib.call("sgc_set_tuning", b"rows_depth", d)
            t = timed(lambda: ops.linear_rows_headmajor_bf16x3(x, wh, wl, sh, 40, 5120, 8), 40)
            t2 = timed(lambda: ops.linear_rows_bf16x3(x, wh, wl, sh, out=y), 40)
            print(f"204800 rows, form {d} (0 staggered, 1 / 2 lockstep depth): head-major {t:.1f} us   row-major {t2:.1f} us", flush=True)
    ops.lib.call("sgc_set_tuning", b"rows_depth", 0)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
