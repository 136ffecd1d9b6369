"""Concurrency of the kernels of a bench run, from a rocprofv3 --kernel-trace CSV: how many kernels run at once, how much of the
wall clock has at least one / at least one MFMA-heavy kernel running, and the CU demand (sum over running kernels of
min(workgroups, 256 * workgroups-per-CU guess)) -- is the chip CU-bound or latency-bound with four scenes in flight?
Usage: python tools/trace_overlap.py kernel_trace.csv"""
import csv, sys, re, collections
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    wgs = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))) * max(1, int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1))))
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    per_cu = 1 if lds > 81920 else 2 if lds > 40960 else 4
    heavy = any(k in name for k in ("conv3d_halo", "conv3d_igemm", "rows_gemm", "level_tail"))
    ev.append((s, 1, min(wgs, 256 * per_cu) / per_cu, heavy, name))
    ev.append((e, -1, min(wgs, 256 * per_cu) / per_cu, heavy, name))
ev.sort(key=lambda t: (t[0], t[1]))
t0, t1 = ev[0][0], ev[-1][0]
# skip the first 40 % (setup, warm-up) and the last 5 %
lo, hi = t0 + (t1 - t0) * 0.55, t0 + (t1 - t0) * 0.9
n = 0; cu = 0.0; nh = 0
hist = collections.Counter(); cuh = collections.Counter(); last = None
busy = heavy_busy = 0; tot = 0; cu_int = 0.0
for t, d, c, h, name in ev:
    if last is not None and lo <= last and t <= hi:
        dt = t - last
        hist[min(n, 8)] += dt; tot += dt
        if n > 0: busy += dt
        if nh > 0: heavy_busy += dt
        cu_int += min(cu, 256.0) * dt
        cuh[min(int(cu // 64), 8)] += dt
    n += d; cu += d * c; nh += d * (1 if h else 0)
    last = t
print("window %.1f ms" % (tot / 1e6))
print("kernels running at once (share of wall): " + "  ".join(f"{k}:{v / tot:.3f}" for k, v in sorted(hist.items())))
print(f"any kernel running {busy / tot:.3f}, an MFMA kernel running {heavy_busy / tot:.3f}")
print(f"mean CU demand (capped at 256) {cu_int / tot:.1f} of 256 = {cu_int / tot / 256:.3f}")
print("CU demand histogram (x64 CUs): " + "  ".join(f"{k * 64}+:{v / tot:.3f}" for k, v in sorted(cuh.items())))
