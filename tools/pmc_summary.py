"""Averages rocprofv3 --pmc counters per dispatch for the kernels whose name contains a substring.
    python tools/pmc_summary.py <dir with *_counter_collection.csv> <kernel substring> [skip_first_n]
Prints one JSON object {counter: mean value per dispatch, "_dispatches": n, "_grid": ..., "_vgpr": ...}."""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
root, pat = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
vals, meta = {}, {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_counter = {}
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if pat not in row["Kernel_Name"]:
                continue
            per_counter.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            meta = dict(_grid=int(row["Grid_Size"]), _wg=int(row["Workgroup_Size"]), _lds=int(row["LDS_Block_Size"]),
                        _vgpr=int(row["VGPR_Count"]), _sgpr=int(row["SGPR_Count"]))
    for name, items in per_counter.items():
        items.sort()
        items = items[skip:]
        if items:
            vals.setdefault(name, []).extend(v for _, v in items)
out = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
out["_dispatches"] = max((len(v) for v in vals.values()), default=0)
out.update(meta)
print(json.dumps(out, indent=1))
