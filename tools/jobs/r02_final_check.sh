#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo bench rc $?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_default.json").readline())
print(d["value"], d["ms_per_step"], d["strict_fp32"]["value"], d["sustained"]["value"], d["self_check"]["mismatching"])
print(json.dumps(d["roofline"])[:300])
cb = d["cpu_baseline"]; print({k: cb[k] for k in ("value", "cores", "sample", "host_cpus", "usable_cores")}, cb.get("one_thread", cb.get("all_cores", {})).get("value"))
PY
