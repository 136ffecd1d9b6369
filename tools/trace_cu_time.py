"""CU-time of a bench run by kernel, from a rocprofv3 --kernel-trace CSV: with four scenes in flight the chip is bound by the sum
of workgroup residency, not by any kernel's latency, so a kernel costs duration x the CUs its workgroups hold.
Residency per CU is estimated from the dispatch record (LDS bytes, registers, workgroup size); a kernel whose workgroups do not
fill the chip is charged only for the CUs it touches.  Usage: python tools/trace_cu_time.py kernel_trace.csv [scenes]"""
import csv, sys, re, collections
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
scenes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
by = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    wg = max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Workgroup_Size_Y"])) * max(1, int(r["Workgroup_Size_Z"]))
    nwg = (int(r["Grid_Size_X"]) * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"]))) // wg
    lds = int(r["LDS_Block_Size"] or 0)
    regs = int(r["VGPR_Count"] or 0) + int(r["Accum_VGPR_Count"] or 0)
    waves = (wg + 63) // 64
    per_simd = 8 if regs <= 64 else 512 // max(regs, 1)
    k = max(1, min(160 * 1024 // lds if lds else 32, max(1, per_simd * 4 // waves), 32 // waves if waves else 32))
    cus = min(256.0, nwg / k)
    e = by[name]
    e[0] += dur * cus / 256.0; e[1] += dur; e[2] += 1
tot = sum(e[0] for e in by.values())
print(f"{'full-chip us/scene':>20s} {'share':>7s} {'kernel us/scene':>16s} {'launches/scene':>15s}  kernel")
for name, (cu, dur, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{cu / scenes:20.1f} {cu / tot:7.3f} {dur / scenes:16.1f} {n / scenes:15.1f}  {name[:90]}")
print(f"total {tot / scenes:.1f} full-chip us per scene over {len(rows)} dispatches")
