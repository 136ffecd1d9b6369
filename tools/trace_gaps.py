"""Where the GPU idles in a single-stream run (the training step): from a rocprofv3 --kernel-trace CSV, the gaps between the end
of everything launched so far and the next kernel's start, attributed to the kernel that FOLLOWS the gap (the one the host was
late with), over the last `window` of the trace (a fraction if <= 1, else milliseconds).  Usage: python tools/trace_gaps.py kernel_trace.csv [steps_in_window] [window]"""
import csv, sys, re, collections
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
k = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:70]) for r in rows))
t0, t1 = k[0][0], max(e for _, e, _ in k)
lo = t1 - ((t1 - t0) * frac if frac <= 1 else frac * 1e6)
k = [x for x in k if x[0] >= lo]
busy_end = k[0][0]
gap_by, gap_n, busy = collections.Counter(), collections.Counter(), 0
big = []
prev = None
for s, e, name in k:
    if s > busy_end:
        g = s - busy_end
        gap_by[name] += g; gap_n[name] += 1
        if g > 20000: big.append((g, prev, name))
        busy_end = s
    busy += max(0, e - busy_end); busy_end = max(busy_end, e)
    prev = name
wall = busy_end - k[0][0]
idle = sum(gap_by.values())
print(f"window {wall / 1e6:.2f} ms = {steps:g} step(s): busy {busy / 1e6 / steps:.2f} ms/step, idle {idle / 1e6 / steps:.2f} ms/step in {sum(gap_n.values()) / steps:.0f} gaps/step; {len(k) / steps:.0f} kernels/step")
print("idle attributed to the kernel after the gap (ms/step, gaps/step):")
for name, g in gap_by.most_common(25):
    print(f"  {g / 1e6 / steps:7.3f}  x{gap_n[name] / steps:6.1f}  {name}")
print("gaps > 20 us (us: after -> before):")
for g, a, b in sorted(big, reverse=True)[:25]:
    print(f"  {g / 1e3:7.1f}  {a}  ->  {b}")
