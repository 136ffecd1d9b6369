#!/bin/bash
# clocks and power while the headline configuration runs (is the throughput regime power-limited?)
mkdir -p gpurun_out
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done ) > gpurun_out/r04_smi_samples.txt &
SMI=$!
sleep 2
timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --sustain 8 > gpurun_out/r04_bench_smi.json 2>/dev/null; echo rc $?
wait $SMI
python - <<'PY'
import json
for ln in open("gpurun_out/r04_smi_samples.txt"):
    ln = ln.strip()
    if not ln.startswith("{"): continue
    try: d = json.loads(ln)
    except Exception: print(ln[:200]); continue
    c = d.get("card0", {})
    print({k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "power", "use", "mclk", "fclk"))})
PY
