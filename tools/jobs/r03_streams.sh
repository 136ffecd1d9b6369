#!/bin/bash
# scenes in flight x hardware queues: throughput of the default workload
for q in 8 16; do
  for s in 4 6 8 12; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 60 --warmup 20 --streams $s --no-cpu-baseline --no-strict-fp32 --sustain 1.5 > gpurun_out/r03_streams_q${q}_s${s}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("gpurun_out/r03_streams_q${q}_s${s}.json"))
print("queues ${q} streams ${s}:", d["value"], "sustained", d["sustained"]["value"], "in flight", d["config"]["scenes_in_flight_per_gpu"], "self_check", d["self_check"]["mismatching"])
PY
  done
done
