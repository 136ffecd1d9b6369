"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_count.py file.s <substring of the symbol>"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
c = collections.Counter()
ops = collections.Counter()
for line in src[start + 1:end]:
    line = line.strip()
    if not line or line[0] in ".;/" or line.endswith(":"):
        continue
    op = line.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    c[kind] += 1
    ops[re.sub(r"_e(32|64)$", "", op)] += 1
print(src[start][:90])
print(dict(c), "total", sum(c.values()))
print(", ".join(f"{k} {v}" for k, v in ops.most_common(40)))
