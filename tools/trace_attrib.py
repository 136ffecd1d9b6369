"""Where does the wall clock of the scenes-in-flight region go?  Reads a rocprofv3 --kernel-trace csv, takes the busiest
window (the graph-replay region: kernels from >= 2 queues/streams overlapping) and attributes every instant of it to the
kernels running at that instant (1 / n each).  Prints per kernel: launches, summed duration, attributed wall share.

    python tools/trace_attrib.py <kernel_trace.csv> [t0_frac t1_frac]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("sgc::", "")
    cut = name.find("(")
    name = name if cut < 0 else name[:cut]
    return name[:70]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", "0"),
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    t_min, t_max = rows[0][0], max(r[1] for r in rows)
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
    w0, w1 = t_min + (t_max - t_min) * f0, t_min + (t_max - t_min) * f1
    rows = [r for r in rows if r[1] > w0 and r[0] < w1]
    events = []
    for i, (s, e, n, st, g) in enumerate(rows):
        events.append((max(s, w0), 1, i))
        events.append((min(e, w1), 0, i))
    events.sort()
    running = set()
    attrib = defaultdict(float)
    idle = 0.0
    last = events[0][0]
    conc_hist = defaultdict(float)
    for t, kind, i in events:
        dt = t - last
        if dt > 0:
            if running:
                for j in running:
                    attrib[rows[j][2]] += dt / len(running)
            else:
                idle += dt
            conc_hist[len(running)] += dt
        last = t
        if kind:
            running.add(i)
        else:
            running.discard(i)
    dur = defaultdict(float)
    cnt = defaultdict(int)
    for s, e, n, st, g in rows:
        dur[n] += e - s
        cnt[n] += 1
    wall = w1 - w0
    print(f"window {wall / 1e6:.2f} ms, {len(rows)} launches, idle {idle / wall * 100:.1f} %, concurrency histogram (share of wall): "
          + ", ".join(f"{k}: {v / wall * 100:.1f}%" for k, v in sorted(conc_hist.items())))
    print(f"{'kernel':70s} {'launches':>8s} {'sum ms':>9s} {'avg us':>8s} {'wall %':>7s}")
    for n, a in sorted(attrib.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{n:70s} {cnt[n]:8d} {dur[n] / 1e6:9.2f} {dur[n] / cnt[n] / 1e3:8.1f} {a / wall * 100:7.2f}")


if __name__ == "__main__":
    main()
